// smhip/back_end.h -- the two other callers of the registrator boundary, restated over the GPU matchers
// (SURVEY.md §8(f) row N4).  Header-only, on top of smhip/registrator.h.
//
//   LoopDetector::CloseLoop        /root/reference/back_end/loop_detector.cc:282-318
//       guess = target_pose^-1 * source_pose with z forced to 0 (:290), a fresh IcpUsingPointMatcher per candidate
//       (:304-307), accept when GetFitnessScore() > accept_scan_match_score and store edge score = -log(score) (:308-316)
//   MapBuilder::SubmapPairMatch    /root/reference/builder/map_builder.cc:399-446
//       matcher from the configured submap_matcher_options (:408-414), guess from the first frames' global poses
//       (:427-429), NormalizeRotation(result) (:434), keep the result when score >= accepted_min_score else the guess
//       (:435-444)
// The scheduling around them (M2DP candidate search, threads, pose graph) is host control plane and stays out.
#ifndef SMHIP_BACK_END_H_
#define SMHIP_BACK_END_H_

#include <cmath>
#include <memory>
#include <utility>
#include <thread>
#include <vector>

#include "smhip/registrator.h"

namespace smhip {
namespace back_end {

using registrator::Matrix4d;
using InnerCloudPtr = data::InnerPointCloudData::Ptr;

inline Matrix4d Multiply(const Matrix4d& a, const Matrix4d& b) {
  Matrix4d r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a(i, k) * b(k, j); r(i, j) = s; }
  return r;
}
// inverse of a rigid transform (what GlobalPose().inverse() is applied to)
inline Matrix4d RigidInverse(const Matrix4d& t) {
  Matrix4d r = Matrix4d::Identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = t(j, i);
  for (int i = 0; i < 3; ++i) r(i, 3) = -(r(i, 0) * t(0, 3) + r(i, 1) * t(1, 3) + r(i, 2) * t(2, 3));
  return r;
}
// common/math.h:240-245: rotation block -> quaternion -> normalised -> rotation block
inline void NormalizeRotation(Matrix4d& t) {
  const double m00 = t(0, 0), m11 = t(1, 1), m22 = t(2, 2), tr = m00 + m11 + m22;
  double w, x, y, z;
  if (tr > 0) { double s = std::sqrt(tr + 1.0); w = 0.5 * s; s = 0.5 / s; x = (t(2, 1) - t(1, 2)) * s; y = (t(0, 2) - t(2, 0)) * s; z = (t(1, 0) - t(0, 1)) * s; }
  else {                                                         // Eigen's Quaternion(Matrix3) branch
    int i = 0; if (m11 > m00) i = 1; if (m22 > t(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(t(i, i) - t(j, j) - t(k, k) + 1.0);
    double q[3]; q[i] = 0.5 * s; s = 0.5 / s;
    w = (t(k, j) - t(j, k)) * s; q[j] = (t(j, i) + t(i, j)) * s; q[k] = (t(k, i) + t(i, k)) * s;
    x = q[0]; y = q[1]; z = q[2];
  }
  const double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  t(0, 0) = 1 - 2 * (y * y + z * z); t(0, 1) = 2 * (x * y - w * z); t(0, 2) = 2 * (x * z + w * y);
  t(1, 0) = 2 * (x * y + w * z); t(1, 1) = 1 - 2 * (x * x + z * z); t(1, 2) = 2 * (y * z - w * x);
  t(2, 0) = 2 * (x * z - w * y); t(2, 1) = 2 * (y * z + w * x); t(2, 2) = 1 - 2 * (x * x + y * y);
}

// back_end/loop_detector.h DetectResult::LoopEdge (the fields CloseLoop touches)
struct LoopEdge {
  std::pair<int, int> close_pair_index{0, 0};   // (target, source)
  Matrix4d init_guess = Matrix4d::Identity();
  Matrix4d transform = Matrix4d::Identity();
  double score = 0.0;
};

struct LoopDetectorSettings {                   // back_end/loop_detector_options.h:39
  float accept_scan_match_score = 0.75f;
  int device = 0;
  int max_points = 1 << 18;          // initial arena size; the matcher grows it to fit the clouds
};

// LoopDetector::CloseLoop for one candidate pair of frames.  `scan_matcher` plays the stack matcher of :304; handing in
// one that outlives the call (the loop detector checks many candidates in a row) keeps its device arena between
// candidates -- it re-sizes itself when a cloud does not fit -- instead of allocating one per candidate.
inline bool CloseLoop(const Matrix4d& target_global_pose, const InnerCloudPtr& target_cloud, const Matrix4d& source_global_pose,
                      const InnerCloudPtr& source_cloud, const LoopDetectorSettings& settings, LoopEdge* edge,
                      registrator::IcpPointMatcherHip* scan_matcher) {
  Matrix4d init_guess = Multiply(RigidInverse(target_global_pose), source_global_pose);     // :287-288
  init_guess(2, 3) = 0.;                                                                     // :290 ("it is a trick")
  edge->init_guess = init_guess;
  scan_matcher->SetInputSource(source_cloud);
  scan_matcher->SetInputTarget(target_cloud);
  scan_matcher->Align(init_guess, edge->transform);
  const double match_score = scan_matcher->GetFitnessScore();
  if (match_score > settings.accept_scan_match_score) {                                     // :309
    edge->score = -std::log(match_score);                                                   // :312
    return true;
  }
  return false;
}
inline bool CloseLoop(const Matrix4d& target_global_pose, const InnerCloudPtr& target_cloud, const Matrix4d& source_global_pose,
                      const InnerCloudPtr& source_cloud, const LoopDetectorSettings& settings, LoopEdge* edge) {
  registrator::IcpPointMatcherHip scan_matcher(settings.device, settings.max_points);       // :304
  scan_matcher.InitWithOptions();
  return CloseLoop(target_global_pose, target_cloud, source_global_pose, source_cloud, settings, edge, &scan_matcher);
}

struct SubmapPairMatchResult {
  Matrix4d transform_to_next = Matrix4d::Identity();   // what SetMatchedTransformedToNext receives
  Matrix4d guess = Matrix4d::Identity();
  double match_score = 0.0;                            // source_submap->match_score_to_previous_submap_
  bool accepted = false;
};

// MapBuilder::SubmapPairMatch for one (source, target) pair of submaps
// `options` = back_end_options.submap_matcher_options (type, accepted_min_score, <registrator_options> text)
inline SubmapPairMatchResult SubmapPairMatch(const registrator::MatcherOptions& options, const InnerCloudPtr& source_submap_cloud,
                                             const Matrix4d& source_first_frame_pose, const InnerCloudPtr& target_submap_cloud,
                                             const Matrix4d& target_first_frame_pose) {
  SubmapPairMatchResult out;
  auto matcher = registrator::CreateMatcher(options, false);                        // :408-414
  SMHIP_CHECK(matcher != nullptr, "CreateMatcher returned null");
  if (matcher->GetType() == registrator::kFastIcp && !target_submap_cloud->GetEigenCloud()->HasNormals())
    target_submap_cloud->CalculateNormals();                                                 // Submap::Cloud() carries normals (submap.cc:161)
  matcher->SetInputSource(source_submap_cloud);
  matcher->SetInputTarget(target_submap_cloud);
  Matrix4d result = Matrix4d::Identity();
  out.guess = Multiply(RigidInverse(target_first_frame_pose), source_first_frame_pose);     // :427-429
  matcher->Align(out.guess, result);
  NormalizeRotation(result);                                                                 // :434
  out.match_score = matcher->GetFitnessScore();
  if (out.match_score >= options.accepted_min_score) { out.transform_to_next = result; out.accepted = true; }   // :437-439
  else { out.transform_to_next = out.guess; out.accepted = false; }                           // :440-444
  return out;
}

// The back end runs up to six SubmapPairMatch tasks at once on its thread pool (map_builder.cc:655, 706-708), each with
// its own matcher.  On the device the same six pairs are ONE batch: `matcher` (from CreateMatcher with the configured
// submap_matcher_options; kept by the caller between batches, so no arena is allocated per pair) aligns all of them in
// one launch sequence through its pair slots.  Matcher types without a batched form fall back to one Align per pair on
// that same matcher.  Results are those of SubmapPairMatch pair by pair.
struct SubmapPairJob {
  InnerCloudPtr source_submap_cloud, target_submap_cloud;
  Matrix4d source_first_frame_pose = Matrix4d::Identity(), target_first_frame_pose = Matrix4d::Identity();
};

inline std::vector<SubmapPairMatchResult> SubmapPairMatchBatch(const registrator::MatcherOptions& options,
                                                               const std::shared_ptr<registrator::Interface>& matcher,
                                                               const std::vector<SubmapPairJob>& jobs) {
  SMHIP_CHECK(matcher != nullptr, "CreateMatcher returned null");
  const size_t K = jobs.size();
  std::vector<SubmapPairMatchResult> out(K);
  if (K == 0) return out;
  std::vector<InnerCloudPtr> src(K), tgt(K);
  std::vector<Matrix4d> guess(K), result;
  std::vector<double> score;
  for (size_t k = 0; k < K; ++k) {
    src[k] = jobs[k].source_submap_cloud; tgt[k] = jobs[k].target_submap_cloud;
    if (matcher->GetType() == registrator::kFastIcp && !tgt[k]->GetEigenCloud()->HasNormals()) tgt[k]->CalculateNormals();   // submap.cc:161
    guess[k] = Multiply(RigidInverse(jobs[k].target_first_frame_pose), jobs[k].source_first_frame_pose);                     // :427-429
    out[k].guess = guess[k];
  }
  bool batched = false;
  if (auto* fast = dynamic_cast<registrator::IcpFastHip*>(matcher.get())) batched = fast->AlignBatch(src, tgt, guess, &result, &score);
  else if (auto* pm = dynamic_cast<registrator::IcpPointMatcherHip*>(matcher.get())) batched = pm->AlignBatch(src, tgt, guess, &result, &score);
  else if (auto* ndt = dynamic_cast<registrator::NdtHip*>(matcher.get())) batched = ndt->AlignBatch(src, tgt, guess, &result, &score);   // lock-step Newton
  else if (auto* ng = dynamic_cast<registrator::NdtGicpHip*>(matcher.get())) batched = ng->AlignBatch(src, tgt, guess, &result, &score);   // lock-step NDT + BFGS
  if (!batched) {
    result.assign(guess.begin(), guess.end());
    score.assign(K, 0.0);
    for (size_t k = 0; k < K; ++k) {
      matcher->SetInputSource(src[k]);
      matcher->SetInputTarget(tgt[k]);
      matcher->Align(guess[k], result[k]);
      score[k] = matcher->GetFitnessScore();
    }
  }
  for (size_t k = 0; k < K; ++k) {
    NormalizeRotation(result[k]);                                                              // :434
    out[k].match_score = score[k];
    if (score[k] >= options.accepted_min_score) { out[k].transform_to_next = result[k]; out[k].accepted = true; }   // :437-439
    else { out[k].transform_to_next = out[k].guess; out[k].accepted = false; }                 // :440-444
  }
  return out;
}

// The reference's own form of concurrency (every matcher also has a batched form above): the back end's thread
// pool runs up to six SubmapPairMatch tasks at once, each with a matcher of its own (map_builder.cc:399-446, 655, 706-708).
// Here: `concurrency` matchers created once and kept between batches (every one owns a device arena and a HIP stream), the
// jobs dealt to them round-robin, one host thread per matcher -- the GPU runs the matchers' streams side by side, so the
// launch latencies, read-backs and host-side Newton / BFGS steps of one pair hide behind the kernels of the others.
// Results are those of SubmapPairMatch pair by pair (every pair runs exactly the single-pair code on its own handle).
class SubmapMatcherPool {
 public:
  SubmapMatcherPool(const registrator::MatcherOptions& options, int concurrency = 6) : options_(options) {
    for (int k = 0; k < std::max(1, concurrency); ++k) {
      auto m = registrator::CreateMatcher(options, false);
      SMHIP_CHECK(m != nullptr, "CreateMatcher returned null");
      // members of a pool align at the same time: IcpFast's single-pair cooperative launch would queue them one behind the other
      if (concurrency > 1 && options.type == registrator::kFastIcp) {
        m->InitWithXml("<param name=\"single_launch\"> 0 </param>");
        m->InitWithOptions();
      }
      matchers_.push_back(m);
    }
  }
  int concurrency() const { return static_cast<int>(matchers_.size()); }
  std::vector<SubmapPairMatchResult> Match(const std::vector<SubmapPairJob>& jobs) {
    const size_t K = jobs.size();
    std::vector<SubmapPairMatchResult> out(K);
    // CalculateNormals mutates a cloud: done up front on this thread, once per cloud (two jobs may share a target)
    if (options_.type == registrator::kFastIcp)
      for (const auto& j : jobs)
        if (!j.target_submap_cloud->GetEigenCloud()->HasNormals()) j.target_submap_cloud->CalculateNormals();
    std::vector<std::thread> workers;
    const size_t C = std::min(matchers_.size(), K);
    for (size_t t = 0; t < C; ++t) {
      workers.emplace_back([&, t] {
        registrator::Interface& m = *matchers_[t];
        for (size_t k = t; k < K; k += C) {
          SubmapPairMatchResult& r = out[k];
          m.SetInputSource(jobs[k].source_submap_cloud);
          m.SetInputTarget(jobs[k].target_submap_cloud);
          Matrix4d result = Matrix4d::Identity();
          r.guess = Multiply(RigidInverse(jobs[k].target_first_frame_pose), jobs[k].source_first_frame_pose);   // :427-429
          m.Align(r.guess, result);
          NormalizeRotation(result);                                                                           // :434
          r.match_score = m.GetFitnessScore();
          if (r.match_score >= options_.accepted_min_score) { r.transform_to_next = result; r.accepted = true; }   // :437-439
          else { r.transform_to_next = r.guess; r.accepted = false; }                                               // :440-444
        }
      });
    }
    for (auto& w : workers) w.join();
    return out;
  }

 private:
  registrator::MatcherOptions options_;
  std::vector<std::shared_ptr<registrator::Interface>> matchers_;
};

}  // namespace back_end
}  // namespace smhip

#endif  // SMHIP_BACK_END_H_
