// smhip/front_end.h -- the front end's caller of the registrator boundary, restated over the GPU matchers.
// Header-only, on top of smhip/registrator.h and smhip/back_end.h (small matrix helpers).
//
//   MapBuilder::ScanMatchProcessing   /root/reference/builder/map_builder.cc:260-397
//       first cloud = first key frame (CalculateNormals for kFastIcp, :286); every later cloud is aligned against the
//       current KEY FRAME with guess = pose_target^-1 * extrapolated pose (:307-309), pose_source = pose_target *
//       align_result (:354); when the motion since the key frame passes the motion filter (translation_range /
//       angle_range, :370-383) the cloud becomes the next key frame (:384-392)
//   PoseExtrapolator (kSimpleCTRV)    /root/reference/builder/pose_extrapolator.cc:90-108, 177-197, 216-240, 296-317
//       constant velocity / turn rate from the oldest and newest pose of a short queue (lidar-only front end,
//       map_builder.cc:72-73)
// Motion compensation (:323-352, "still in test" there), submap insertion and threading stay out: control plane.
#ifndef SMHIP_FRONT_END_H_
#define SMHIP_FRONT_END_H_

#include <cmath>
#include <deque>
#include <memory>
#include <utility>

#include "smhip/back_end.h"
#include "smhip/registrator.h"

namespace smhip {
namespace front_end {

using registrator::Matrix4d;
using back_end::Multiply;
using back_end::NormalizeRotation;
using back_end::RigidInverse;
using InnerCloudPtr = data::InnerPointCloudData::Ptr;

// common/math.h:108-127 (x, y, z Euler angles of a rotation matrix)
inline void RotationMatrixToEulerAngles(const Matrix4d& R, double e[3]) {
  const double sy = std::sqrt(R(0, 0) * R(0, 0) + R(1, 0) * R(1, 0));
  if (!(sy < 1e-6)) { e[0] = std::atan2(R(2, 1), R(2, 2)); e[1] = std::atan2(-R(2, 0), sy); e[2] = std::atan2(R(1, 0), R(0, 0)); }
  else { e[0] = std::atan2(-R(1, 2), R(1, 1)); e[1] = std::atan2(-R(2, 0), sy); e[2] = 0; }
}
// common/math.h:130-138: Rz(e2) * Ry(e1) * Rx(e0)
inline Matrix4d EulerAnglesToRotation(const double e[3]) {
  const double cx = std::cos(e[0]), sx = std::sin(e[0]), cy = std::cos(e[1]), sy = std::sin(e[1]), cz = std::cos(e[2]), sz = std::sin(e[2]);
  Matrix4d R = Matrix4d::Identity();
  R(0, 0) = cz * cy; R(0, 1) = cz * sy * sx - sz * cx; R(0, 2) = cz * sy * cx + sz * sx;
  R(1, 0) = sz * cy; R(1, 1) = sz * sy * sx + cz * cx; R(1, 2) = sz * sy * cx - cz * sx;
  R(2, 0) = -sy;     R(2, 1) = cy * sx;                R(2, 2) = cy * cx;
  return R;
}

// PoseExtrapolator in Mode::kSimpleCTRV
class PoseExtrapolatorCTRV {
 public:
  explicit PoseExtrapolatorCTRV(double pose_queue_duration_s = 0.001) : duration_(pose_queue_duration_s) {}   // map_builder.cc:49
  bool Empty() const { return queue_.empty(); }
  double GetLastPoseTime() const { return queue_.empty() ? 0.0 : queue_.back().first; }
  void AddPose(double time, const Matrix4d& pose) {                         // pose_extrapolator.cc:90-108
    queue_.emplace_back(time, pose);
    if (queue_.size() == 1u) return;
    while (queue_.size() > 2 && queue_[1].first <= time - duration_) queue_.pop_front();
    UpdateVelocitiesFromPoses();
  }
  void InitRoughLinearVelocity(double vx, double vy, double vz) { linear_[0] = vx; linear_[1] = vy; linear_[2] = vz; }   // :210-214
  Matrix4d ExtrapolatePose(double time) const {                             // :177-197
    const Matrix4d& newest = queue_.back().second;
    const double dt = time - queue_.back().first;
    const double de[3] = {angular_[0] * dt, angular_[1] * dt, angular_[2] * dt};
    Matrix4d out = Multiply(RotationOnly(newest), EulerAnglesToRotation(de));   // newest rotation * delta (:185-188)
    for (int i = 0; i < 3; ++i) out(i, 3) = newest(i, 3) + dt * linear_[i];    // :183-184, :311-317
    return out;
  }

 private:
  static Matrix4d RotationOnly(const Matrix4d& t) { Matrix4d r = t; r(0, 3) = r(1, 3) = r(2, 3) = 0; return r; }
  void UpdateVelocitiesFromPoses() {                                         // :216-240
    if (queue_.size() < 2) return;
    const double delta = queue_.back().first - queue_.front().first;
    if (delta < duration_) return;
    const Matrix4d& a = queue_.front().second;
    const Matrix4d& b = queue_.back().second;
    for (int i = 0; i < 3; ++i) linear_[i] = (b(i, 3) - a(i, 3)) / delta;
    double e[3];
    RotationMatrixToEulerAngles(Multiply(RigidInverse(RotationOnly(a)), RotationOnly(b)), e);
    for (int i = 0; i < 3; ++i) angular_[i] = e[i] / delta;
  }
  double duration_;
  std::deque<std::pair<double, Matrix4d>> queue_;
  double linear_[3] = {0, 0, 0}, angular_[3] = {0, 0, 0};
};

struct MotionFilter {                      // builder/map_builder.h:79-82
  float translation_range = 0.35f;
  float angle_range = 1.5f;                // degrees, sum of |Euler angles|
};

struct FrameResult {
  Matrix4d pose = Matrix4d::Identity();    // pose_source: the scan in the map frame
  Matrix4d guess = Matrix4d::Identity();
  Matrix4d align_result = Matrix4d::Identity();
  double score = 1.0;
  bool matched = false;                    // false for the first cloud (and while the extrapolator initialises)
  bool new_key_frame = false;
};

// MapBuilder::ScanMatchProcessing, one cloud per call
class ScanMatcherFrontEnd {
 public:
  // device_target_prep (IcpFastHip only): the key frame stays resident on the GPU.  Its normals are computed there
  // (SetInputTargetRaw / PromoteSourceToTarget) instead of the host CalculateNormals of map_builder.cc:286,389, the
  // target is not re-sent with every scan (the reference calls SetInputTarget per scan, :317), and a scan that becomes a
  // key frame is handed over from the source slot without a second upload.  false = the reference's call sequence.
  ScanMatcherFrontEnd(std::shared_ptr<registrator::Interface> scan_matcher, const MotionFilter& filter, bool use_extrapolator = true,
                      bool device_target_prep = false)
      : scan_matcher_(std::move(scan_matcher)), filter_(filter), use_extrapolator_(use_extrapolator) {
    if (device_target_prep) device_icp_ = dynamic_cast<registrator::IcpFastHip*>(scan_matcher_.get());
  }
  bool DeviceTargetPrep() const { return device_icp_ != nullptr; }

  PoseExtrapolatorCTRV& Extrapolator() { return extrapolator_; }

  FrameResult ProcessCloud(const InnerCloudPtr& source_cloud, double source_time) {
    FrameResult out;
    if (!got_first_point_cloud_) {                                          // :280-293
      got_first_point_cloud_ = true;
      target_cloud_ = source_cloud;
      if (device_icp_) device_icp_->SetInputTargetRaw(target_cloud_);
      else if (scan_matcher_->GetType() == registrator::kFastIcp) target_cloud_->CalculateNormals();
      if (use_extrapolator_) extrapolator_.AddPose(source_time, Matrix4d::Identity());
      out.new_key_frame = true;
      return out;
    }
    if (use_extrapolator_ && source_time < extrapolator_.GetLastPoseTime()) {  // :296-300
      target_cloud_ = source_cloud;
      if (device_icp_) device_icp_->SetInputTargetRaw(target_cloud_);
      return out;
    }
    Matrix4d pose_source = pose_target_;
    if (use_extrapolator_) pose_source = extrapolator_.ExtrapolatePose(source_time);   // :302-305
    Matrix4d guess = Multiply(RigidInverse(pose_target_), pose_source);        // :307
    NormalizeRotation(guess);                                                  // :308
    Matrix4d align_result = Matrix4d::Identity();
    if (!device_icp_) scan_matcher_->SetInputTarget(target_cloud_);            // :317 (resident otherwise)
    scan_matcher_->SetInputSource(source_cloud);                               // :329
    scan_matcher_->Align(guess, align_result);                                 // :333
    pose_source = Multiply(pose_target_, align_result);                        // :354
    accumulative_transform_ = align_result;                                    // :355
    if (use_extrapolator_) extrapolator_.AddPose(source_time, pose_source);    // :357
    out.pose = pose_source; out.guess = guess; out.align_result = align_result;
    out.score = scan_matcher_->GetFitnessScore(); out.matched = true;
    const double tx = accumulative_transform_(0, 3), ty = accumulative_transform_(1, 3), tz = accumulative_transform_(2, 3);
    const float accu_translation = static_cast<float>(std::sqrt(tx * tx + ty * ty + tz * tz));   // :370-371
    double e[3];
    RotationMatrixToEulerAngles(accumulative_transform_, e);
    const float accu_angles = static_cast<float>((std::fabs(e[0]) + std::fabs(e[1]) + std::fabs(e[2])) * (180. / M_PI));   // :374-377
    if (accu_translation >= filter_.translation_range || (filter_.angle_range > 1e-3 && accu_angles >= filter_.angle_range)) {   // :379-383
      accumulative_transform_ = Matrix4d::Identity();
      target_cloud_ = source_cloud;
      if (device_icp_) device_icp_->PromoteSourceToTarget();
      else if (scan_matcher_->GetType() == registrator::kFastIcp) target_cloud_->CalculateNormals();   // :389
      pose_target_ = pose_source;
      out.new_key_frame = true;
    }
    return out;
  }

 private:
  std::shared_ptr<registrator::Interface> scan_matcher_;
  MotionFilter filter_;
  bool use_extrapolator_;
  registrator::IcpFastHip* device_icp_ = nullptr;
  PoseExtrapolatorCTRV extrapolator_;
  bool got_first_point_cloud_ = false;
  InnerCloudPtr target_cloud_;
  Matrix4d pose_target_ = Matrix4d::Identity();
  Matrix4d accumulative_transform_ = Matrix4d::Identity();
};

}  // namespace front_end
}  // namespace smhip

#endif  // SMHIP_FRONT_END_H_
