// smhip/registrator.h -- C++ host-side mirror of static_map::registrator::Interface for the
// MI355X backend.  Same class shape, method names, option names and error conventions as
//   /root/reference/registrators/interface.h:41-128   (Type, MatcherOptions, Interface, REG_ macro)
//   /root/reference/registrators/interface.cc:38-173  (SetInput*, InitWithXml, PrintOptions, CreateMatcher)
//   /root/reference/registrators/icp_fast.h:38-64     (IcpFast and its options_)
// but free of Eigen / PCL / glog / pugixml so it compiles anywhere; every bit of arithmetic happens
// behind the C ABI of include/smhip.h.  INTEGRATION.md shows how the class plugs into the
// reference's CreateMatcher switch; an Eigen adapter is provided when <Eigen/Core> is available.
//
// Header-only; link with -lsmhip.
#ifndef SMHIP_REGISTRATOR_H_
#define SMHIP_REGISTRATOR_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../smhip.h"

#if defined(SMHIP_REGISTRATOR_THROW_ON_CHECK)
#include <stdexcept>
#define SMHIP_CHECK(cond, msg)                                         \
  do {                                                                 \
    if (!(cond)) throw std::runtime_error(std::string("CHECK failed: ") + (msg)); \
  } while (0)
#else
// glog CHECK semantics of the reference (e.g. interface.cc:66-67, icp_fast.cc:422-430): abort
#define SMHIP_CHECK(cond, msg)                                         \
  do {                                                                 \
    if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", (msg), __FILE__, __LINE__); std::abort(); } \
  } while (0)
#endif

namespace smhip {
namespace data {

// builder/data/cloud_types.h:46-56
struct InnerPointType {
  float x = 0.f, y = 0.f, z = 0.f, intensity = 0.f, factor = 0.f;
};

// The parts of data::EigenPointCloud (cloud_types.h:121-147) the registrators use:
// 3xN column-major double points / normals (xyzxyz...).
struct EigenPointCloud {
  std::vector<double> points;    // 3 * N
  std::vector<double> normals;   // 3 * N or empty
  int size() const { return static_cast<int>(points.size() / 3); }
  bool HasNormals() const { return !normals.empty(); }   // cloud_types.cc:304

  // cloud_types.cc:328-345
  void FromPointCloud(const std::vector<InnerPointType>& inner_points) {
    SMHIP_CHECK(!inner_points.empty(), "empty cloud");
    points.resize(3 * inner_points.size());
    normals.clear();
    for (size_t i = 0; i < inner_points.size(); ++i) {
      points[3 * i] = inner_points[i].x; points[3 * i + 1] = inner_points[i].y; points[3 * i + 2] = inner_points[i].z;
    }
  }
  // cloud_types.cc:347-368: kd-box subsampling + least-squares normals (runs in libsmhip.so, host side)
  void CalculateNormals() {
    const int n = size();
    SMHIP_CHECK(n > 0, "CalculateNormals on an empty cloud");
    std::vector<double> op(3 * (size_t)n), on(3 * (size_t)n);
    int m = 0;
    const smhip_status s = smhip_calculate_normals_f64(points.data(), n, op.data(), on.data(), &m);
    SMHIP_CHECK(s == SMHIP_OK, "smhip_calculate_normals_f64 failed");
    op.resize(3 * (size_t)m); on.resize(3 * (size_t)m);
    points.swap(op); normals.swap(on);
  }
};

// The parts of data::InnerPointCloudData (cloud_types.h:153-195) the registrators use.
class InnerPointCloudData {
 public:
  using Ptr = std::shared_ptr<InnerPointCloudData>;
  explicit InnerPointCloudData(const std::vector<InnerPointType>& cloud) : inner_cloud_(cloud) {
    eigen_cloud_.reset(new EigenPointCloud);
    if (!cloud.empty()) eigen_cloud_->FromPointCloud(cloud);
  }
  bool Empty() const { return inner_cloud_.empty(); }
  void CalculateNormals() { eigen_cloud_->CalculateNormals(); }
  std::shared_ptr<EigenPointCloud> GetEigenCloud() const { return eigen_cloud_; }
  const std::vector<InnerPointType>& GetInnerCloud() const { return inner_cloud_; }

 private:
  std::vector<InnerPointType> inner_cloud_;
  std::shared_ptr<EigenPointCloud> eigen_cloud_;
};

}  // namespace data

namespace registrator {

// Column-major 4x4 double, the storage of Eigen::Matrix4d.
struct Matrix4d {
  double m[16];
  static Matrix4d Identity() { Matrix4d r; std::memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r; }
  double& operator()(int r, int c) { return m[4 * c + r]; }
  double operator()(int r, int c) const { return m[4 * c + r]; }
  double* data() { return m; }
  const double* data() const { return m; }
};

// interface.h:41-50 -- same numeric values, so config files keep working
enum Type { kNoType, kIcpPM, kLibicp, kNdtWithGicp, kLegoLoam, kNdt, kFastIcp, kTypeCount };

enum class OptionItemDataType : uint8_t { kInt32, kFloat32, kBool };   // interface.h:52

struct InnerOptionItem {   // interface.h:54-57
  OptionItemDataType data_type;
  void* data_ptr = nullptr;
};

// interface.h:59-65.  pugi::xml_node is replaced by the text of the matching
// <registrator_options type="N"> element (what map_builder_options.cc:44-55 selects).
struct MatcherOptions {
  Type type = kIcpPM;
  float accepted_min_score = 0.7f;
  std::string registrator_options_node;   // "<param name=...>v</param>..." ; empty = defaults
  std::string inner_filters_node;
  int device = 0;                          // new: which GPU the matcher lives on
};

class Interface {
 public:
  using InnerCloudPtr = data::InnerPointCloudData::Ptr;

  Interface() = default;
  virtual ~Interface() {}
  Interface(const Interface&) = delete;
  Interface& operator=(const Interface&) = delete;

  // interface.cc:62-90.  Accepts the children of a <registrator_options> element:
  //   <param name="max_iteration"> 100 </param>
  void InitWithXml(const std::string& node) {
    size_t pos = 0;
    while ((pos = node.find("<param", pos)) != std::string::npos) {
      const size_t name_at = node.find("name", pos);
      const size_t q0 = node.find_first_of("\"'", name_at);
      const size_t q1 = node.find(node[q0], q0 + 1);
      const size_t gt = node.find('>', q1);
      const size_t close = node.find("</param>", gt);
      SMHIP_CHECK(name_at != std::string::npos && q0 != std::string::npos && q1 != std::string::npos &&
                      gt != std::string::npos && close != std::string::npos, "malformed <param> element");
      const std::string param_name = node.substr(q0 + 1, q1 - q0 - 1);
      const std::string text = node.substr(gt + 1, close - gt - 1);
      SMHIP_CHECK(inner_options_.count(param_name) > 0, "Init an unknown option of this matcher!");   // :66-67
      const InnerOptionItem& item = inner_options_.at(param_name);
      switch (item.data_type) {
        case OptionItemDataType::kInt32: *reinterpret_cast<int32_t*>(item.data_ptr) = std::atoi(text.c_str()); break;
        case OptionItemDataType::kFloat32: *reinterpret_cast<float*>(item.data_ptr) = static_cast<float>(std::atof(text.c_str())); break;
        case OptionItemDataType::kBool: {
          const size_t b = text.find_first_not_of(" \t\r\n");
          const char c = b == std::string::npos ? '0' : text[b];
          *reinterpret_cast<bool*>(item.data_ptr) = (c == '1' || c == 't' || c == 'T' || c == 'y' || c == 'Y');   // pugixml as_bool
          break;
        }
      }
      pos = close;
    }
  }
  void InitInnerFiltersWithXml(const std::string&) {}          // interface.cc:92-111: a stub in the reference too
  void EnableInnerCompensation() { inner_compensation_ = true; }    // interface.cc:34
  void DisableInnerCompensation() { inner_compensation_ = false; }  // interface.cc:36
  virtual void InitWithOptions() {}                                 // interface.h:94
  void PrintOptions() {                                             // interface.cc:113-137
    for (const auto& kv : inner_options_) {
      std::cout << std::setw(25) << kv.first << " -> ";
      switch (kv.second.data_type) {
        case OptionItemDataType::kInt32: std::cout << *reinterpret_cast<int32_t*>(kv.second.data_ptr); break;
        case OptionItemDataType::kFloat32: std::cout << std::setprecision(6) << *reinterpret_cast<float*>(kv.second.data_ptr); break;
        case OptionItemDataType::kBool: std::cout << std::boolalpha << *reinterpret_cast<bool*>(kv.second.data_ptr); break;
      }
      std::cout << std::endl;
    }
  }

  virtual void SetInputSource(InnerCloudPtr source_cloud) {         // interface.cc:38-48
    if (!source_cloud) { source_cloud_ = nullptr; return; }
    if (source_cloud->Empty()) { std::fprintf(stderr, "[WARNING] cloud is empty.\n"); return; }
    source_cloud_ = source_cloud;
  }
  virtual void SetInputTarget(InnerCloudPtr target_cloud) {         // interface.cc:50-60
    if (!target_cloud) { target_cloud_ = nullptr; return; }
    if (target_cloud->Empty()) { std::fprintf(stderr, "[WARNING] cloud is empty.\n"); return; }
    target_cloud_ = target_cloud;
  }
  virtual double GetFitnessScore() { return final_score_; }         // interface.h:100
  virtual bool Align(const Matrix4d& guess, Matrix4d& result) = 0;  // interface.h:103-104
  virtual Type GetType() const { return type_; }                    // interface.h:106

 protected:
  double final_score_ = 0.0;
  Type type_ = kNoType;
  InnerCloudPtr source_cloud_ = nullptr;
  InnerCloudPtr target_cloud_ = nullptr;
  std::unordered_map<std::string, InnerOptionItem> inner_options_;
  bool inner_compensation_ = false;
};

#define SMHIP_REG_REGISTRATOR_INNER_OPTION(NAME, TYPE, VARIABLE) \
  this->inner_options_[NAME].data_type = TYPE;                   \
  this->inner_options_[NAME].data_ptr = &VARIABLE;

// GPU replacement of registrator::IcpFast (icp_fast.h:38-64, icp_fast.cc:407-529).
class IcpFastHip : public Interface {
 public:
  explicit IcpFastHip(int device = 0, int max_points = 1 << 18) : device_(device), max_points_(max_points) {
    this->type_ = kFastIcp;
    // the three names icp_fast.cc:407-419 registers, plus the backend's own knobs
    SMHIP_REG_REGISTRATOR_INNER_OPTION("knn_normal_estimate", OptionItemDataType::kInt32, options_.knn_for_normal_estimate);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("max_iteration", OptionItemDataType::kInt32, options_.max_iteration);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("dist_outlier_ratio", OptionItemDataType::kFloat32, options_.dist_outlier_ratio);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("device_id", OptionItemDataType::kInt32, device_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("nn_mode", OptionItemDataType::kInt32, options_.nn_mode);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("grid_cell", OptionItemDataType::kFloat32, options_.grid_cell);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("exact_matches", OptionItemDataType::kBool, options_.exact_matches);
  }
  ~IcpFastHip() override { if (handle_) smhip_destroy(handle_); }

  void InitWithOptions() override { EnsureHandle(); }

  void SetInputSource(InnerCloudPtr cloud) override {               // icp_fast.cc:421-425
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    EnsureHandle();
    const auto& e = *cloud->GetEigenCloud();
    Check(smhip_set_source_f64(handle_, 0, e.points.data(), e.size()), "smhip_set_source_f64");
  }
  void SetInputTarget(InnerCloudPtr cloud) override {               // icp_fast.cc:427-431
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    SMHIP_CHECK(cloud->GetEigenCloud()->HasNormals(), "CHECK(cloud->GetEigenCloud()->HasNormals())");
    EnsureHandle();
    const auto& e = *cloud->GetEigenCloud();
    Check(smhip_set_target_f64(handle_, 0, e.points.data(), e.normals.data(), e.size()), "smhip_set_target_f64");
  }
  bool Align(const Matrix4d& guess, Matrix4d& result) override {    // icp_fast.cc:455-529
    EnsureHandle();
    double score = 0.0;
    const smhip_status s = smhip_icp_align(handle_, guess.data(), result.data(), &score, &stats_);
    if (s != SMHIP_OK) {
      // the reference would CHECK-abort on these (icp_fast.cc:81,113); here: PRINT_ERROR + false
      std::fprintf(stderr, "[ERROR] IcpFastHip::Align: %s (%s)\n", smhip_status_string(s), smhip_last_error(handle_));
      result = guess;
      return false;
    }
    this->final_score_ = score;
    return true;                                                    // icp_fast.cc:528: always true
  }
  const smhip_icp_stats& LastStats() const { return stats_; }

  // Device-resident target preparation (the backend's own additions; the reference has the caller run
  // EigenPointCloud::CalculateNormals on the host, map_builder.cc:286,389, before SetInputTarget).
  // The raw cloud is uploaded and its normals are computed on the GPU (csrc/prep_normals.hip); returns the number of
  // target points that kept a normal.  The host cloud is left untouched (it gets no normals).
  int SetInputTargetRaw(InnerCloudPtr cloud) {
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    EnsureHandle();
    const auto& e = *cloud->GetEigenCloud();
    std::vector<float> rows(static_cast<size_t>(e.size()) * 3);
    const double* p = e.points.data();
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = static_cast<float>(p[i]);   // 3xN column-major == N rows of xyz
    int n_out = 0;
    Check(smhip_prepare_target_f32(handle_, 0, rows.data(), 3, e.size(), &n_out), "smhip_prepare_target_f32");
    return n_out;
  }
  // The source already resident from the last SetInputSource becomes the target (the key-frame hand-over of
  // map_builder.cc:379-392) without a second upload.
  int PromoteSourceToTarget() {
    EnsureHandle();
    int n_out = 0;
    Check(smhip_prepare_target_from_source(handle_, 0, 0, &n_out), "smhip_prepare_target_from_source");
    return n_out;
  }

 private:
  void Check(smhip_status s, const char* what) {
    if (s != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] %s: %s (%s)\n", what, smhip_status_string(s), handle_ ? smhip_last_error(handle_) : "");
      SMHIP_CHECK(false, what);
    }
  }
  void EnsureHandle() {
    if (!handle_) {
      const smhip_status s = smhip_create(device_, nullptr, 1, max_points_, max_points_, &handle_);
      if (s != SMHIP_OK) { std::fprintf(stderr, "[ERROR] smhip_create: %s\n", smhip_status_string(s)); }
      SMHIP_CHECK(s == SMHIP_OK, "no usable MI355X (gfx950) device: there is no CPU fallback");
    }
    smhip_icp_options o;
    smhip_icp_default_options(&o);
    o.max_iteration = options_.max_iteration;
    o.dist_outlier_ratio = options_.dist_outlier_ratio;
    o.nn_mode = options_.nn_mode;
    o.grid_cell = options_.grid_cell;
    o.exact_matches = options_.exact_matches ? 1 : 0;
    Check(smhip_icp_set_options(handle_, &o), "smhip_icp_set_options");
  }

  struct {
    int32_t knn_for_normal_estimate = 7;     // icp_fast.h:57 (unused there as well)
    int32_t max_iteration = 100;             // icp_fast.h:58
    float dist_outlier_ratio = 0.7f;         // icp_fast.h:59
    int32_t nn_mode = SMHIP_NN_GRID;
    float grid_cell = 0.25f;
    bool exact_matches = false;
  } options_;
  int32_t device_ = 0;
  int max_points_;
  smhip_handle handle_ = nullptr;
  smhip_icp_stats stats_{};
};

// GPU replacement of registrator::Ndt (ndt.h / ndt.cc:29-64): pclomp NDT with resolution 1.0 and the
// KDTREE neighbourhood; score = pcl getFitnessScore() (mean squared 1-NN distance, lower is better).
class NdtHip : public Interface {
 public:
  explicit NdtHip(int device = 0, int max_source = 1 << 18, int max_target = 1 << 21)
      : device_(device), max_source_(max_source), max_target_(max_target) {
    this->type_ = kNdt;
    smhip_ndt_default_options(&opt_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("device_id", OptionItemDataType::kInt32, device_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("resolution", OptionItemDataType::kFloat32, opt_.resolution);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("step_size", OptionItemDataType::kFloat32, opt_.step_size);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("max_iterations", OptionItemDataType::kInt32, opt_.max_iterations);
  }
  ~NdtHip() override { if (handle_) smhip_destroy(handle_); }
  void InitWithOptions() override { EnsureHandle(); }

  bool Align(const Matrix4d& guess, Matrix4d& result) override {      // ndt.cc:38-64
    if (!this->source_cloud_ || !this->target_cloud_) return false;   // :40-42
    EnsureHandle();
    // ToPclPointCloud of both clouds on every Align (:44-51): the 20-byte InnerPointType AoS goes up as is
    const auto& s = this->source_cloud_->GetInnerCloud();
    const auto& t = this->target_cloud_->GetInnerCloud();
    if (smhip_set_source_f32(handle_, 0, &s[0].x, 5, static_cast<int>(s.size())) != SMHIP_OK ||
        smhip_set_target_f32(handle_, 0, &t[0].x, 5, nullptr, 0, static_cast<int>(t.size())) != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtHip: %s\n", smhip_last_error(handle_));
      return false;
    }
    double score = 0.0;
    const smhip_status st = smhip_ndt_align(handle_, guess.data(), result.data(), &score, &stats_);
    if (st != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtHip::Align: %s (%s)\n", smhip_status_string(st), smhip_last_error(handle_));
      result = guess;
      return false;
    }
    this->final_score_ = score;                                       // :60
    return true;
  }
  const smhip_ndt_stats& LastStats() const { return stats_; }

 private:
  void EnsureHandle() {
    if (!handle_) {
      const smhip_status s = smhip_create(device_, nullptr, 1, max_source_, max_target_, &handle_);
      SMHIP_CHECK(s == SMHIP_OK, "no usable MI355X (gfx950) device: there is no CPU fallback");
    }
    SMHIP_CHECK(smhip_ndt_set_options(handle_, &opt_) == SMHIP_OK, "smhip_ndt_set_options");
  }
  smhip_ndt_options opt_;
  int32_t device_ = 0;
  int max_source_, max_target_;
  smhip_handle handle_ = nullptr;
  smhip_ndt_stats stats_{};
};

// GPU replacement of registrator::NdtWithGicp (ndt_gicp.h:39-77, ndt_gicp.cc:28-112): ApproximateVoxelGrid on both
// clouds -> pcl NDT -> pcl GICP; score = exp(-GICP fitness), Align returns false (result = guess) when the NDT
// fitness is > 1.  Option names are the reference's (ndt_gicp.cc:31-36).
class NdtGicpHip : public Interface {
 public:
  explicit NdtGicpHip(int device = 0, int max_source = 1 << 18, int max_target = 1 << 21)
      : device_(device), max_source_(max_source), max_target_(max_target) {
    this->type_ = kNdtWithGicp;
    smhip_ndt_gicp_default_options(&opt_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("device_id", OptionItemDataType::kInt32, device_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("use_ndt", OptionItemDataType::kBool, use_ndt_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("using_voxel_filter", OptionItemDataType::kBool, using_voxel_filter_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("voxel_resolution", OptionItemDataType::kFloat32, opt_.voxel_resolution);
  }
  ~NdtGicpHip() override { if (handle_) smhip_destroy(handle_); }
  void InitWithOptions() override { EnsureHandle(); }

  bool Align(const Matrix4d& guess, Matrix4d& result) override {      // ndt_gicp.cc:55-112
    if (!this->source_cloud_ || !this->target_cloud_) return false;
    EnsureHandle();
    // ToPclPointCloud of both stored clouds on every Align (:59-76)
    const auto& s = this->source_cloud_->GetInnerCloud();
    const auto& t = this->target_cloud_->GetInnerCloud();
    if (smhip_ndt_gicp_set_source_f32(handle_, &s[0].x, 5, static_cast<int>(s.size())) != SMHIP_OK ||
        smhip_ndt_gicp_set_target_f32(handle_, &t[0].x, 5, static_cast<int>(t.size())) != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtGicpHip: %s\n", smhip_last_error(handle_));
      return false;
    }
    double score = 0.0;
    const smhip_status st = smhip_ndt_gicp_align(handle_, guess.data(), result.data(), &score, &stats_);
    if (st != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtGicpHip::Align: %s (%s)\n", smhip_status_string(st), smhip_last_error(handle_));
      result = guess;
      return false;
    }
    this->final_score_ = score;                                       // :102 / :107
    return stats_.ok != 0;
  }
  const smhip_ndt_gicp_stats& LastStats() const { return stats_; }

 private:
  void EnsureHandle() {
    if (!handle_) {
      const smhip_status s = smhip_create(device_, nullptr, 2, max_source_, max_target_ > max_source_ ? max_target_ : max_source_, &handle_);
      SMHIP_CHECK(s == SMHIP_OK, "no usable MI355X (gfx950) device: there is no CPU fallback");
    }
    opt_.use_ndt = use_ndt_ ? 1 : 0;
    opt_.using_voxel_filter = using_voxel_filter_ ? 1 : 0;
    SMHIP_CHECK(smhip_ndt_gicp_set_options(handle_, &opt_) == SMHIP_OK, "smhip_ndt_gicp_set_options");
  }
  smhip_ndt_gicp_options opt_;
  bool use_ndt_ = true, using_voxel_filter_ = true;
  int32_t device_ = 0;
  int max_source_, max_target_;
  smhip_handle handle_ = nullptr;
  smhip_ndt_gicp_stats stats_{};
};

// GPU replacement of registrator::IcpUsingPointMatcher (icp_pointmatcher.cc:104-247): the
// libpointmatcher chain RandomSampling(0.9) -> SamplingSurfaceNormal(knn 7, method 1) -> KDTree(1-NN)
// -> TrimmedDist(0.7) -> PointToPlane -> Counter(150) + Differential(1e-3, 1e-2, 4), followed by the
// post-hoc score over the full reading against the raw reference (:112-143); Align returns score >= 0.6.
class IcpPointMatcherHip : public Interface {
 public:
  explicit IcpPointMatcherHip(int device = 0, int max_points = 1 << 18) : device_(device), max_points_(max_points) {
    this->type_ = kIcpPM;
    SMHIP_REG_REGISTRATOR_INNER_OPTION("device_id", OptionItemDataType::kInt32, device_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("random_sampling_prob", OptionItemDataType::kFloat32, prob_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("random_seed", OptionItemDataType::kInt32, seed_);
  }
  ~IcpPointMatcherHip() override { if (handle_) smhip_destroy(handle_); }
  void InitWithOptions() override { EnsureHandle(); }

  void SetInputSource(InnerCloudPtr cloud) override {               // icp_pointmatcher.cc:84-92
    if (!cloud || cloud->Empty()) { std::fprintf(stderr, "[ERROR] Empty cloud.\n"); return; }
    reading_ = DropNan(cloud->GetInnerCloud());
  }
  void SetInputTarget(InnerCloudPtr cloud) override {               // :94-102
    if (!cloud || cloud->Empty()) { std::fprintf(stderr, "[ERROR] Empty cloud.\n"); return; }
    reference_ = DropNan(cloud->GetInnerCloud());
  }
  bool Align(const Matrix4d& guess, Matrix4d& result) override {    // :104-149
    if (reading_.empty() || reference_.empty()) return false;
    EnsureHandle();
    // reading filter: RandomSampling (seeded here; the reference uses std::rand())
    std::vector<float> sampled;
    sampled.reserve(reading_.size());
    uint32_t state = static_cast<uint32_t>(seed_) * 2654435761u + 12345u;
    for (size_t i = 0; i + 2 < reading_.size(); i += 3) {
      state = state * 1664525u + 1013904223u;
      const float r = static_cast<float>(state >> 8) * (1.0f / 16777216.0f);
      if (prob_ >= 1.0f || r < prob_) { sampled.push_back(reading_[i]); sampled.push_back(reading_[i + 1]); sampled.push_back(reading_[i + 2]); }
    }
    // reference filter: SamplingSurfaceNormal == CalculateNormals
    data::EigenPointCloud ref;
    ref.points.assign(reference_.begin(), reference_.end());
    ref.CalculateNormals();
    smhip_icp_options o; smhip_icp_default_options(&o);
    o.max_iteration = 150; o.dist_outlier_ratio = 0.7f; o.early_exit = 1;
    if (smhip_icp_set_options(handle_, &o) != SMHIP_OK ||
        smhip_set_source_f32(handle_, 0, sampled.data(), 3, static_cast<int>(sampled.size() / 3)) != SMHIP_OK ||
        smhip_set_target_f64(handle_, 0, ref.points.data(), ref.normals.data(), ref.size()) != SMHIP_OK) return Fail(guess, result);
    double score = 0.0;
    if (smhip_icp_align(handle_, guess.data(), result.data(), &score, &stats_) != SMHIP_OK) return Fail(guess, result);
    // final score: the FULL reading transformed by `result` against the RAW reference, one trimmed pass (:112-143)
    std::vector<double> raw(reference_.begin(), reference_.end()), up(raw.size(), 0.0);
    for (size_t i = 2; i < up.size(); i += 3) up[i] = 1.0;
    o.max_iteration = 1; o.early_exit = 0;
    Matrix4d ignored;
    if (smhip_icp_set_options(handle_, &o) != SMHIP_OK ||
        smhip_set_source_f32(handle_, 0, reading_.data(), 3, static_cast<int>(reading_.size() / 3)) != SMHIP_OK ||
        smhip_set_target_f64(handle_, 0, raw.data(), up.data(), static_cast<int>(raw.size() / 3)) != SMHIP_OK ||
        smhip_icp_align(handle_, result.data(), ignored.data(), &score, nullptr) != SMHIP_OK) return Fail(guess, result);
    this->final_score_ = score;                                      // :143
    return this->final_score_ >= 0.6;                                // :145-148
  }

 private:
  static std::vector<float> DropNan(const std::vector<data::InnerPointType>& in) {   // InnerCloudToPmPoints, :43-71
    std::vector<float> out;
    out.reserve(3 * in.size());
    for (const auto& p : in)
      if (!(p.x != p.x) && !(p.y != p.y) && !(p.z != p.z)) { out.push_back(p.x); out.push_back(p.y); out.push_back(p.z); }
    return out;
  }
  bool Fail(const Matrix4d& guess, Matrix4d& result) {
    std::fprintf(stderr, "[ERROR] IcpPointMatcherHip::Align: %s\n", smhip_last_error(handle_));
    result = guess;
    return false;
  }
  void EnsureHandle() {
    if (!handle_) {
      const smhip_status s = smhip_create(device_, nullptr, 1, max_points_, max_points_, &handle_);
      SMHIP_CHECK(s == SMHIP_OK, "no usable MI355X (gfx950) device: there is no CPU fallback");
    }
  }
  std::vector<float> reading_, reference_;     // xyz of the non-NaN points
  float prob_ = 0.9f;                           // icp_pointmatcher.cc:172
  int32_t seed_ = 0;
  int32_t device_ = 0;
  int max_points_;
  smhip_handle handle_ = nullptr;
  smhip_icp_stats stats_{};
};

// interface.cc:139-173.  kFastIcp selects the HIP matcher; the matchers that have no HIP
// implementation yet report "Wrong type" exactly like an unknown enum value does there.
inline std::shared_ptr<Interface> CreateMatcher(const MatcherOptions& options, bool verbose = false) {
  std::shared_ptr<Interface> matcher;
  switch (options.type) {
    case kFastIcp:
      matcher.reset(new IcpFastHip(options.device));
      break;
    case kNdt:
      matcher.reset(new NdtHip(options.device));
      break;
    case kIcpPM:
      matcher.reset(new IcpPointMatcherHip(options.device));
      break;
    case kNdtWithGicp:
      matcher.reset(new NdtGicpHip(options.device));
      break;
    default:
      std::fprintf(stderr, "[ERROR] Wrong type\n");
      return nullptr;
  }
  if (!options.registrator_options_node.empty()) matcher->InitWithXml(options.registrator_options_node);
  if (verbose) matcher->PrintOptions();
  matcher->InitWithOptions();
  return matcher;
}

}  // namespace registrator
}  // namespace smhip

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
namespace smhip {
namespace registrator {
// Eigen::Matrix4d is column-major, the same 16 doubles.
inline bool Align(Interface& m, const Eigen::Matrix4d& guess, Eigen::Matrix4d& result) {
  Matrix4d g, r;
  std::memcpy(g.m, guess.data(), sizeof(g.m));
  const bool ok = m.Align(g, r);
  std::memcpy(result.data(), r.m, sizeof(r.m));
  return ok;
}
}  // namespace registrator
}  // namespace smhip
#endif
#endif

#endif  // SMHIP_REGISTRATOR_H_
