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

#include <algorithm>
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
  void InitWithXml(const std::string& node_in) {
    // comments are not parameters (pugixml skips them): blank them out, keeping offsets
    std::string node = node_in;
    for (size_t c = node.find("<!--"); c != std::string::npos; c = node.find("<!--", c)) {
      const size_t e = node.find("-->", c + 4);
      const size_t stop = e == std::string::npos ? node.size() : e + 3;
      for (size_t k = c; k < stop; ++k) node[k] = ' ';
      c = stop;
    }
    size_t pos = 0;
    while ((pos = node.find("<param", pos)) != std::string::npos) {
      // every search is bounded by this element's own start tag
      const size_t gt = node.find('>', pos);
      SMHIP_CHECK(gt != std::string::npos, "malformed <param> element");
      const size_t name_at = node.find("name", pos);
      SMHIP_CHECK(name_at != std::string::npos && name_at < gt, "malformed <param> element: no name attribute");
      const size_t q0 = node.find_first_of("\"'", name_at);
      SMHIP_CHECK(q0 != std::string::npos && q0 < gt, "malformed <param> element: unquoted name");
      const size_t q1 = node.find(node[q0], q0 + 1);
      SMHIP_CHECK(q1 != std::string::npos && q1 < gt, "malformed <param> element: unterminated name");
      const size_t close = node.find("</param>", gt);
      SMHIP_CHECK(close != std::string::npos, "malformed <param> element: no </param>");
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
  // interface.cc:34-36.  Only IcpFast reads the flag (icp_fast.cc:487-489, 282-287), and what it selects there is
  // EigenPointCloud::ApplyMotionCompensation (cloud_types.cc:306-321), whose per-point transform is declared as
  //     const Eigen::Matrix4d transform = common::InterpolateTransform(I, transform, factors[i]);
  // -- the new `transform` shadows the argument inside its own initialiser, so the branch interpolates towards an
  // uninitialised matrix: undefined behaviour, and nothing in the reference ever calls the setter.  There is no defined
  // result to reproduce, so IcpFastHip::Align / AlignBatch REFUSE to run with the flag set (error + false) instead of quietly
  // running the uncompensated iteration.
  void EnableInnerCompensation() { inner_compensation_ = true; }
  void DisableInnerCompensation() { inner_compensation_ = false; }
  bool InnerCompensationEnabled() const { return inner_compensation_; }
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

// The handle's device arena is sized at creation (no hipMalloc inside Align).  The reference has no size limit, so
// the adapters below size the handle from the clouds they are handed: when a cloud does not fit, the handle is
// destroyed and re-created with room to spare (x1.5, so a growing submap does not re-create it on every call) and
// the adapter re-uploads what the old handle held.  "max_points" (inner option) is only the INITIAL capacity.
struct DeviceArena {
  smhip_handle handle = nullptr;
  int device = 0, slots = 1, cap_source = 0, cap_target = 0;
  static constexpr int kHardMaxSource = 4194304;     // smhip_create's limit on max_source_points
  ~DeviceArena() { if (handle) smhip_destroy(handle); }
  DeviceArena() = default;
  DeviceArena(const DeviceArena&) = delete;
  DeviceArena& operator=(const DeviceArena&) = delete;
  static int Grow(int need, int have) {
    if (need <= have) return have;
    long long g = static_cast<long long>(need) + need / 2;
    g = (g + 65535) / 65536 * 65536;
    return static_cast<int>(g > 0x7fff0000LL ? 0x7fff0000LL : g);
  }
  // Makes sure a handle with room for ns source and nt target points exists on `dev`.
  // *recreated = true when a new handle replaced an old one (its device-side clouds are gone).
  // Returns false (and prints why) when the device refuses: no gfx950 device, out of memory, ns beyond the hard limit.
  bool Reserve(int dev, int nslots, int ns, int nt, bool* recreated = nullptr) {
    if (recreated) *recreated = false;
    if (handle && dev == device && nslots <= slots && ns <= cap_source && nt <= cap_target) return true;
    int want_s = Grow(ns, handle ? cap_source : 0), want_t = Grow(nt, handle ? cap_target : 0);
    if (want_s > kHardMaxSource) want_s = kHardMaxSource;
    if (ns > kHardMaxSource) {
      std::fprintf(stderr, "[ERROR] source cloud of %d points exceeds the backend's limit of %d\n", ns, kHardMaxSource);
      return false;
    }
    if (want_s < 1) want_s = 1;
    if (want_t < 1) want_t = 1;
    const bool had = handle != nullptr;
    if (handle) { smhip_destroy(handle); handle = nullptr; }
    const smhip_status s = smhip_create(dev, nullptr, nslots, want_s, want_t, &handle);
    if (s != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] smhip_create(device %d, %d slot(s), %d / %d points): %s\n", dev, nslots, want_s, want_t, smhip_status_string(s));
      handle = nullptr; cap_source = cap_target = 0;
      return false;
    }
    device = dev; slots = nslots; cap_source = want_s; cap_target = want_t;
    if (recreated) *recreated = had;
    return true;
  }
};

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
    SMHIP_REG_REGISTRATOR_INNER_OPTION("max_points", OptionItemDataType::kInt32, max_points_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("nn_mode", OptionItemDataType::kInt32, options_.nn_mode);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("nn_epsilon", OptionItemDataType::kFloat32, options_.nn_epsilon);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("grid_cell", OptionItemDataType::kFloat32, options_.grid_cell);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("exact_matches", OptionItemDataType::kBool, options_.exact_matches);
    // true (default): a single Align is ONE cooperative launch (csrc/icp_one.hip) -- the front end's one-pair-at-a-time call.  Matchers
    // that align at the same time from several threads (the back end's pool of six) set it false: cooperative launches of several
    // handles run one after the other on the device's cooperative queue (six threads: 583 Aligns/s against 781 as separate launches).
    SMHIP_REG_REGISTRATOR_INNER_OPTION("single_launch", OptionItemDataType::kBool, options_.single_launch);
  }

  void InitWithOptions() override { EnsureHandle(0, 0, true, true); }

  void SetInputSource(InnerCloudPtr cloud) override {               // icp_fast.cc:421-425
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    // the cloud the device holds changes only when the upload succeeds; a refused cloud leaves the matcher without a
    // source (Align then fails loudly instead of matching the previous scan)
    source_ok_ = EnsureHandle(cloud->GetEigenCloud()->size(), 0, false, true) && UploadSource(*cloud);
    source_keep_ = source_ok_ ? cloud : nullptr;
  }
  void SetInputTarget(InnerCloudPtr cloud) override {               // icp_fast.cc:427-431
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    SMHIP_CHECK(cloud->GetEigenCloud()->HasNormals(), "CHECK(cloud->GetEigenCloud()->HasNormals())");
    SetTarget(cloud, kTargetWithNormals);
  }
  bool Align(const Matrix4d& guess, Matrix4d& result) override {    // icp_fast.cc:455-529
    if (RefuseInnerCompensation()) { result = guess; return false; }
    if (!source_ok_ || !target_ok_ || !EnsureHandle(0, 0, true, true)) {   // a cloud the device could not take: not an abort
      std::fprintf(stderr, "[ERROR] IcpFastHip::Align: the input clouds are not on the device\n");
      result = guess;
      return false;
    }
    double score = 0.0;
    const smhip_status s = smhip_icp_align(arena_.handle, guess.data(), result.data(), &score, &stats_);
    if (s != SMHIP_OK) {
      // the reference would CHECK-abort on these (icp_fast.cc:81,113); here: PRINT_ERROR + false
      std::fprintf(stderr, "[ERROR] IcpFastHip::Align: %s (%s)\n", smhip_status_string(s), smhip_last_error(arena_.handle));
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
  // target points that kept a normal (0 when the device could not take the cloud).  The host cloud is left untouched.
  int SetInputTargetRaw(InnerCloudPtr cloud) {
    SMHIP_CHECK(cloud != nullptr, "CHECK(cloud)");
    SMHIP_CHECK(cloud->GetEigenCloud() != nullptr, "CHECK(cloud->GetEigenCloud())");
    return SetTarget(cloud, kTargetRaw) ? prepared_points_ : 0;
  }
  // The source already resident from the last SetInputSource becomes the target (the key-frame hand-over of
  // map_builder.cc:379-392) without a second upload.
  int PromoteSourceToTarget() {
    if (!source_keep_ || !source_ok_ || !EnsureHandle(0, source_keep_->GetEigenCloud()->size() / 4 + 8, true, true)) return 0;
    int n_out = 0;
    if (!Ok(smhip_prepare_target_from_source(arena_.handle, 0, 0, &n_out), "smhip_prepare_target_from_source")) {
      // the slot's target arrays may have been overwritten before the failure: no target until the next SetInputTarget
      target_ok_ = false; target_keep_ = nullptr; target_kind_ = kNoTarget;
      return 0;
    }
    target_keep_ = source_keep_; target_kind_ = kTargetRaw; prepared_points_ = n_out; target_ok_ = true;
    return n_out;
  }
  int CapacitySource() const { return arena_.cap_source; }
  int CapacityTarget() const { return arena_.cap_target; }

  // K independent (source, target) pairs in ONE launch sequence through K pair slots of a second handle that this
  // matcher keeps between calls -- the six concurrent SubmapPairMatch tasks of the back end (map_builder.cc:655,
  // 706-708) as one batch instead of six matchers with an arena each.  Targets must carry normals.
  // results / scores are resized to K; returns false (results = guesses) when the device refused a cloud or a pair failed.
  bool AlignBatch(const std::vector<InnerCloudPtr>& sources, const std::vector<InnerCloudPtr>& targets,
                  const std::vector<Matrix4d>& guesses, std::vector<Matrix4d>* results, std::vector<double>* scores,
                  std::vector<smhip_icp_stats>* stats = nullptr) {
    const int K = static_cast<int>(sources.size());
    SMHIP_CHECK(K > 0 && targets.size() == sources.size() && guesses.size() == sources.size() && results && scores, "AlignBatch: sizes");
    results->assign(guesses.begin(), guesses.end());
    scores->assign(K, 0.0);
    if (RefuseInnerCompensation()) return false;
    int ns = 1, nt = 1;
    for (int k = 0; k < K; ++k) {
      SMHIP_CHECK(sources[k] && targets[k] && sources[k]->GetEigenCloud() && targets[k]->GetEigenCloud(), "CHECK(cloud)");
      SMHIP_CHECK(targets[k]->GetEigenCloud()->HasNormals(), "CHECK(cloud->GetEigenCloud()->HasNormals())");
      ns = std::max(ns, sources[k]->GetEigenCloud()->size());
      nt = std::max(nt, targets[k]->GetEigenCloud()->size());
    }
    if (!batch_arena_.Reserve(device_, std::max(K, batch_arena_.slots), ns, nt)) return false;
    smhip_handle h = batch_arena_.handle;
    if (!ApplyOptions(h)) return false;
    for (int k = 0; k < K; ++k) {
      const auto& es = *sources[k]->GetEigenCloud();
      const auto& et = *targets[k]->GetEigenCloud();
      if (smhip_set_source_f64(h, k, es.points.data(), es.size()) != SMHIP_OK ||
          smhip_set_target_f64(h, k, et.points.data(), et.normals.data(), et.size()) != SMHIP_OK) {
        std::fprintf(stderr, "[ERROR] IcpFastHip::AlignBatch: %s\n", smhip_last_error(h));
        return false;
      }
    }
    std::vector<double> g(16 * static_cast<size_t>(K)), r(16 * static_cast<size_t>(K));
    std::vector<smhip_icp_stats> st(K);
    for (int k = 0; k < K; ++k) std::memcpy(&g[16 * static_cast<size_t>(k)], guesses[k].data(), sizeof(double) * 16);
    const smhip_status s = smhip_icp_align_batch(h, K, g.data(), r.data(), scores->data(), st.data());
    if (s != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] IcpFastHip::AlignBatch: %s (%s)\n", smhip_status_string(s), smhip_last_error(h));
      scores->assign(K, 0.0);
      return false;
    }
    for (int k = 0; k < K; ++k) std::memcpy((*results)[k].data(), &r[16 * static_cast<size_t>(k)], sizeof(double) * 16);
    if (stats) *stats = st;
    return true;
  }

 private:
  enum TargetKind { kNoTarget, kTargetWithNormals, kTargetRaw };
  // see Interface::EnableInnerCompensation: the reference's compensated branch has no defined result
  bool RefuseInnerCompensation() const {
    if (!this->inner_compensation_) return false;
    std::fprintf(stderr, "[ERROR] IcpFastHip: inner compensation is enabled, and the reference's branch for it (icp_fast.cc:487-489 -> "
                         "cloud_types.cc:306-321) reads an uninitialised transform: no defined result to reproduce. Call DisableInnerCompensation().\n");
    return true;
  }
  bool Ok(smhip_status s, const char* what) {
    if (s == SMHIP_OK) return true;
    std::fprintf(stderr, "[ERROR] %s: %s (%s)\n", what, smhip_status_string(s), arena_.handle ? smhip_last_error(arena_.handle) : "");
    return false;
  }
  // SetInputTarget / SetInputTargetRaw: what the device holds (target_keep_, target_kind_) changes only on success.  A refused
  // cloud (NaN / Inf, too large for the device) leaves the matcher WITHOUT a target: the slot may be half-written and the
  // previous key frame must not be matched silently (Align returns false), nor re-uploaded after a re-size.
  bool SetTarget(const InnerCloudPtr& cloud, TargetKind kind) {
    target_ok_ = EnsureHandle(0, cloud->GetEigenCloud()->size(), true, false) && UploadTarget(*cloud, kind);
    target_keep_ = target_ok_ ? cloud : nullptr;
    target_kind_ = target_ok_ ? kind : kNoTarget;
    return target_ok_;
  }
  bool UploadSource(const data::InnerPointCloudData& cloud) {
    const auto& e = *cloud.GetEigenCloud();
    return Ok(smhip_set_source_f64(arena_.handle, 0, e.points.data(), e.size()), "smhip_set_source_f64");
  }
  bool UploadTarget(const data::InnerPointCloudData& cloud, TargetKind kind) {
    const auto& e = *cloud.GetEigenCloud();
    if (kind == kTargetWithNormals)
      return Ok(smhip_set_target_f64(arena_.handle, 0, e.points.data(), e.normals.data(), e.size()), "smhip_set_target_f64");
    std::vector<float> rows(static_cast<size_t>(e.size()) * 3);
    const double* p = e.points.data();
    for (size_t i = 0; i < rows.size(); ++i) rows[i] = static_cast<float>(p[i]);   // 3xN column-major == N rows of xyz
    return Ok(smhip_prepare_target_f32(arena_.handle, 0, rows.data(), 3, e.size(), &prepared_points_), "smhip_prepare_target_f32");
  }
  bool ApplyOptions(smhip_handle h) {
    smhip_icp_options o;
    smhip_icp_default_options(&o);
    o.max_iteration = options_.max_iteration;
    o.dist_outlier_ratio = options_.dist_outlier_ratio;
    o.nn_mode = options_.nn_mode;
    o.nn_epsilon = options_.nn_epsilon;
    o.grid_cell = options_.grid_cell;
    o.exact_matches = options_.exact_matches ? 1 : 0;
    o.no_single_kernel = options_.single_launch ? 0 : 1;
    return Ok(smhip_icp_set_options(h, &o), "smhip_icp_set_options");
  }
  // Handle with room for ns / nt points (0 = whatever it has).  After a re-creation the clouds the old handle held
  // are uploaded again from the shared_ptrs kept for that purpose, except the one the caller is about to upload itself.
  bool EnsureHandle(int ns, int nt, bool restore_source, bool restore_target) {
    const bool first = arena_.handle == nullptr;
    const int init = max_points_ > 0 ? max_points_ : 1;
    bool recreated = false;
    if (!arena_.Reserve(device_, 1, std::max(ns, first ? init : 0), std::max(nt, first ? init : 0), &recreated)) {
      SMHIP_CHECK(!first || smhip_device_count() > 0, "no usable MI355X (gfx950) device: there is no CPU fallback");
      return false;
    }
    if (!ApplyOptions(arena_.handle)) return false;
    if (recreated) {
      if (restore_source && source_keep_ && !UploadSource(*source_keep_)) { source_ok_ = false; return false; }
      if (restore_target && target_keep_ && target_kind_ != kNoTarget && !UploadTarget(*target_keep_, target_kind_)) { target_ok_ = false; return false; }
    }
    return true;
  }

  struct {
    int32_t knn_for_normal_estimate = 7;     // icp_fast.h:57 (unused there as well)
    int32_t max_iteration = 100;             // icp_fast.h:58
    float dist_outlier_ratio = 0.7f;         // icp_fast.h:59
    int32_t nn_mode = SMHIP_NN_GRID;         // SMHIP_NN_NABO = libnabo's tree and epsilon search (what the reference runs)
    float nn_epsilon = 3.16f;                // icp_fast.cc:174
    float grid_cell = 0.25f;
    bool exact_matches = false;
    bool single_launch = true;
  } options_;
  int32_t device_ = 0;
  int32_t max_points_;
  DeviceArena arena_;
  DeviceArena batch_arena_;                    // AlignBatch's K-slot handle
  InnerCloudPtr source_keep_, target_keep_;   // what the device holds (icp_fast.cc:424,431 deep-copies; here: to re-upload after a re-size)
  TargetKind target_kind_ = kNoTarget;
  int prepared_points_ = 0;
  bool source_ok_ = false, target_ok_ = false;   // the device holds the cloud of the last SetInputSource / SetInputTarget
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
  void InitWithOptions() override { EnsureHandle(0, 0); }

  bool Align(const Matrix4d& guess, Matrix4d& result) override {      // ndt.cc:38-64
    if (!this->source_cloud_ || !this->target_cloud_) return false;   // :40-42
    // ToPclPointCloud of both clouds on every Align (:44-51): the 20-byte InnerPointType AoS goes up as is
    const auto& s = this->source_cloud_->GetInnerCloud();
    const auto& t = this->target_cloud_->GetInnerCloud();
    if (!EnsureHandle(static_cast<int>(s.size()), static_cast<int>(t.size()))) { result = guess; return false; }
    smhip_handle handle_ = arena_.handle;
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

  // K independent (source, target) pairs as ONE lock-step batch through K pair slots of a second handle this matcher keeps
  // between calls (smhip_ndt_align_batch): the back end's concurrent SubmapPairMatch tasks (map_builder.cc:399-446, 655) with
  // the Ndt matcher.  Each pair's result is what Align gives for it, bit for bit.  results / scores are resized to K; false
  // (results = guesses) when the device refused a cloud.
  bool AlignBatch(const std::vector<InnerCloudPtr>& sources, const std::vector<InnerCloudPtr>& targets,
                  const std::vector<Matrix4d>& guesses, std::vector<Matrix4d>* results, std::vector<double>* scores,
                  std::vector<smhip_ndt_stats>* stats = nullptr) {
    const int K = static_cast<int>(sources.size());
    SMHIP_CHECK(K > 0 && targets.size() == sources.size() && guesses.size() == sources.size() && results && scores, "AlignBatch: sizes");
    results->assign(guesses.begin(), guesses.end());
    scores->assign(K, 0.0);
    int ns = 1, nt = 1;
    for (int k = 0; k < K; ++k) {
      if (!sources[k] || !targets[k] || sources[k]->Empty() || targets[k]->Empty()) return false;      // ndt.cc:40-42
      ns = std::max(ns, static_cast<int>(sources[k]->GetInnerCloud().size()));
      nt = std::max(nt, static_cast<int>(targets[k]->GetInnerCloud().size()));
    }
    if (!batch_arena_.Reserve(device_, std::max(K, batch_arena_.slots), ns, nt)) return false;
    smhip_handle h = batch_arena_.handle;
    SMHIP_CHECK(smhip_ndt_set_options(h, &opt_) == SMHIP_OK, "smhip_ndt_set_options");
    for (int k = 0; k < K; ++k) {
      const auto& s = sources[k]->GetInnerCloud();
      const auto& t = targets[k]->GetInnerCloud();
      if (smhip_set_source_f32(h, k, &s[0].x, 5, static_cast<int>(s.size())) != SMHIP_OK ||
          smhip_set_target_f32(h, k, &t[0].x, 5, nullptr, 0, static_cast<int>(t.size())) != SMHIP_OK) {
        std::fprintf(stderr, "[ERROR] NdtHip::AlignBatch: %s\n", smhip_last_error(h));
        return false;
      }
    }
    std::vector<double> g(16 * static_cast<size_t>(K)), r(16 * static_cast<size_t>(K));
    std::vector<smhip_ndt_stats> st(K);
    for (int k = 0; k < K; ++k) std::memcpy(&g[16 * static_cast<size_t>(k)], guesses[k].data(), sizeof(double) * 16);
    const smhip_status rc = smhip_ndt_align_batch(h, 0, K, g.data(), r.data(), scores->data(), st.data());
    if (rc != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtHip::AlignBatch: %s (%s)\n", smhip_status_string(rc), smhip_last_error(h));
      scores->assign(K, 0.0);
      return false;
    }
    for (int k = 0; k < K; ++k) std::memcpy((*results)[k].data(), &r[16 * static_cast<size_t>(k)], sizeof(double) * 16);
    if (stats) *stats = st;
    return true;
  }

 private:
  bool EnsureHandle(int ns, int nt) {
    const bool first = arena_.handle == nullptr;
    if (!arena_.Reserve(device_, 1, std::max(ns, first ? max_source_ : 0), std::max(nt, first ? max_target_ : 0))) {
      SMHIP_CHECK(!first || smhip_device_count() > 0, "no usable MI355X (gfx950) device: there is no CPU fallback");
      return false;
    }
    SMHIP_CHECK(smhip_ndt_set_options(arena_.handle, &opt_) == SMHIP_OK, "smhip_ndt_set_options");
    return true;
  }
  smhip_ndt_options opt_;
  int32_t device_ = 0;
  int max_source_, max_target_;
  DeviceArena arena_;
  DeviceArena batch_arena_;                    // AlignBatch's K-slot handle
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
  void InitWithOptions() override { EnsureHandle(0, 0); }

  bool Align(const Matrix4d& guess, Matrix4d& result) override {      // ndt_gicp.cc:55-112
    if (!this->source_cloud_ || !this->target_cloud_) return false;
    // ToPclPointCloud of both stored clouds on every Align (:59-76)
    const auto& s = this->source_cloud_->GetInnerCloud();
    const auto& t = this->target_cloud_->GetInnerCloud();
    if (!EnsureHandle(static_cast<int>(s.size()), static_cast<int>(t.size()))) { result = guess; return false; }
    smhip_handle handle_ = arena_.handle;
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

  // K independent (source, target) pairs as ONE lock-step batch through K jobs of a second handle this matcher keeps between
  // calls (smhip_ndt_gicp_align_batch): the back end's concurrent SubmapPairMatch tasks (map_builder.cc:399-446, 655) with
  // the NdtWithGicp matcher.  Each pair's result and score are what Align gives for it, bit for bit; (*ok)[k] is Align's
  // return value for pair k (false with result = guess when its NDT fitness is > 1, ndt_gicp.cc:105-108).  false when the
  // device refused a cloud (results = guesses).
  bool AlignBatch(const std::vector<InnerCloudPtr>& sources, const std::vector<InnerCloudPtr>& targets,
                  const std::vector<Matrix4d>& guesses, std::vector<Matrix4d>* results, std::vector<double>* scores,
                  std::vector<char>* ok = nullptr, std::vector<smhip_ndt_gicp_stats>* stats = nullptr) {
    const int K = static_cast<int>(sources.size());
    SMHIP_CHECK(K > 0 && targets.size() == sources.size() && guesses.size() == sources.size() && results && scores, "AlignBatch: sizes");
    results->assign(guesses.begin(), guesses.end());
    scores->assign(K, 0.0);
    if (ok) ok->assign(K, 0);
    int ns = 1, nt = 1;
    for (int k = 0; k < K; ++k) {
      if (!sources[k] || !targets[k] || sources[k]->Empty() || targets[k]->Empty()) return false;
      ns = std::max(ns, static_cast<int>(sources[k]->GetInnerCloud().size()));
      nt = std::max(nt, static_cast<int>(targets[k]->GetInnerCloud().size()));
    }
    // two pair slots per job; the target side also has to hold the down-sampled source
    if (!batch_arena_.Reserve(device_, std::max(2 * K, batch_arena_.slots), ns, std::max(nt, ns))) return false;
    smhip_handle h = batch_arena_.handle;
    opt_.use_ndt = use_ndt_ ? 1 : 0;
    opt_.using_voxel_filter = using_voxel_filter_ ? 1 : 0;
    SMHIP_CHECK(smhip_ndt_gicp_set_options(h, &opt_) == SMHIP_OK, "smhip_ndt_gicp_set_options");
    for (int k = 0; k < K; ++k) {
      const auto& s = sources[k]->GetInnerCloud();
      const auto& t = targets[k]->GetInnerCloud();
      if (smhip_ndt_gicp_set_source_f32_job(h, k, &s[0].x, 5, static_cast<int>(s.size())) != SMHIP_OK ||
          smhip_ndt_gicp_set_target_f32_job(h, k, &t[0].x, 5, static_cast<int>(t.size())) != SMHIP_OK) {
        std::fprintf(stderr, "[ERROR] NdtGicpHip::AlignBatch: %s\n", smhip_last_error(h));
        return false;
      }
    }
    std::vector<double> g(16 * static_cast<size_t>(K)), r(16 * static_cast<size_t>(K));
    std::vector<smhip_ndt_gicp_stats> st(K);
    for (int k = 0; k < K; ++k) std::memcpy(&g[16 * static_cast<size_t>(k)], guesses[k].data(), sizeof(double) * 16);
    const smhip_status rc = smhip_ndt_gicp_align_batch(h, 0, K, g.data(), r.data(), scores->data(), st.data());
    if (rc != SMHIP_OK) {
      std::fprintf(stderr, "[ERROR] NdtGicpHip::AlignBatch: %s (%s)\n", smhip_status_string(rc), smhip_last_error(h));
      scores->assign(K, 0.0);
      return false;
    }
    for (int k = 0; k < K; ++k) {
      std::memcpy((*results)[k].data(), &r[16 * static_cast<size_t>(k)], sizeof(double) * 16);
      if (ok) (*ok)[k] = st[k].ok != 0;
    }
    if (stats) *stats = st;
    return true;
  }

 private:
  // two pair slots (working space of the matcher); the target side also has to hold the down-sampled source
  bool EnsureHandle(int ns, int nt) {
    const bool first = arena_.handle == nullptr;
    const int want_s = std::max(ns, first ? max_source_ : 0);
    const int want_t = std::max(std::max(nt, ns), first ? std::max(max_target_, max_source_) : 0);
    if (!arena_.Reserve(device_, 2, want_s, want_t)) {
      SMHIP_CHECK(!first || smhip_device_count() > 0, "no usable MI355X (gfx950) device: there is no CPU fallback");
      return false;
    }
    opt_.use_ndt = use_ndt_ ? 1 : 0;
    opt_.using_voxel_filter = using_voxel_filter_ ? 1 : 0;
    SMHIP_CHECK(smhip_ndt_gicp_set_options(arena_.handle, &opt_) == SMHIP_OK, "smhip_ndt_gicp_set_options");
    return true;
  }
  smhip_ndt_gicp_options opt_;
  bool use_ndt_ = true, using_voxel_filter_ = true;
  int32_t device_ = 0;
  int max_source_, max_target_;
  DeviceArena arena_;
  DeviceArena batch_arena_;                    // AlignBatch's handle: two pair slots per job
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
    SMHIP_REG_REGISTRATOR_INNER_OPTION("max_points", OptionItemDataType::kInt32, max_points_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("random_sampling_prob", OptionItemDataType::kFloat32, prob_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("random_seed", OptionItemDataType::kInt32, seed_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("nn_mode", OptionItemDataType::kInt32, nn_mode_);
    SMHIP_REG_REGISTRATOR_INNER_OPTION("nn_epsilon", OptionItemDataType::kFloat32, nn_epsilon_);
  }
  void InitWithOptions() override { EnsureHandle(0, 0); }

  void SetInputSource(InnerCloudPtr cloud) override {               // icp_pointmatcher.cc:84-92
    if (!cloud || cloud->Empty()) { std::fprintf(stderr, "[ERROR] Empty cloud.\n"); return; }
    reading_ = DropNan(cloud->GetInnerCloud());
    reading_on_device_ = false;
  }
  void SetInputTarget(InnerCloudPtr cloud) override {               // :94-102
    if (!cloud || cloud->Empty()) { std::fprintf(stderr, "[ERROR] Empty cloud.\n"); return; }
    reference_ = DropNan(cloud->GetInnerCloud());
    reference_on_device_ = false;
  }
  // Slot 0 is the ICP pair (sampled reading vs CalculateNormals(reference)), slot 1 the score pair (full reading vs
  // raw reference).  Each cloud is uploaded once; sampling, normals, the 150-iteration loop and the score pass all
  // run on the device.
  bool Align(const Matrix4d& guess, Matrix4d& result) override {    // :104-149
    if (reading_.empty() || reference_.empty()) return false;
    const int nr = static_cast<int>(reading_.size() / 3), nf = static_cast<int>(reference_.size() / 3);
    bool recreated = false;
    if (!EnsureHandle(nr, nf, &recreated)) { result = guess; return false; }
    if (recreated) reading_on_device_ = reference_on_device_ = false;
    smhip_handle h = arena_.handle;
    if (!reading_on_device_) {
      if (smhip_set_source_f32(h, 1, reading_.data(), 3, nr) != SMHIP_OK) return Fail(guess, result);
      reading_on_device_ = true;
    }
    if (!reference_on_device_) {
      if (smhip_set_target_f32(h, 1, reference_.data(), 3, nullptr, 0, nf) != SMHIP_OK) return Fail(guess, result);
      reference_on_device_ = true;
    }
    int n_sampled = 0, n_target = 0;
    // reading filter: RandomSampling(prob) (:170-174); reference filter: SamplingSurfaceNormal == CalculateNormals (:176-184)
    if (smhip_sample_source(h, 1, 0, prob_, static_cast<uint32_t>(seed_), &n_sampled) != SMHIP_OK ||
        smhip_prepare_target_from_target(h, 1, 0, &n_target) != SMHIP_OK) return Fail(guess, result);
    smhip_icp_options o; smhip_icp_default_options(&o);
    o.max_iteration = 150; o.dist_outlier_ratio = 0.7f; o.early_exit = 1;           // :196-224
    o.nn_mode = nn_mode_; o.nn_epsilon = nn_epsilon_;                               // KDTreeMatcher knn 1, epsilon 3.16 (:186-191) when nn_mode = 2
    if (smhip_icp_set_options(h, &o) != SMHIP_OK) return Fail(guess, result);
    double score = 0.0;
    if (smhip_icp_align(h, guess.data(), result.data(), &score, &stats_) != SMHIP_OK) return Fail(guess, result);
    // final score: the FULL reading transformed by `result` against the RAW reference, one trimmed pass (:112-143)
    if (smhip_icp_trimmed_score(h, 1, result.data(), 0.7f, &score, nullptr) != SMHIP_OK) return Fail(guess, result);
    this->final_score_ = score;                                      // :143
    return this->final_score_ >= 0.6;                                // :145-148
  }
  const smhip_icp_stats& LastStats() const { return stats_; }

  // K independent (reading, reference) pairs through 2 K pair slots of a handle kept between calls: slots [0, K) hold the
  // ICP pairs, slots [K, 2 K) the score pairs; ONE launch sequence runs the K 150-iteration loops together.
  // accepted[k] = score >= 0.6 (Align's bool); results = guesses for pairs the device could not take.
  bool AlignBatch(const std::vector<InnerCloudPtr>& sources, const std::vector<InnerCloudPtr>& targets,
                  const std::vector<Matrix4d>& guesses, std::vector<Matrix4d>* results, std::vector<double>* scores,
                  std::vector<bool>* accepted = nullptr) {
    const int K = static_cast<int>(sources.size());
    SMHIP_CHECK(K > 0 && targets.size() == sources.size() && guesses.size() == sources.size() && results && scores, "AlignBatch: sizes");
    results->assign(guesses.begin(), guesses.end());
    scores->assign(K, 0.0);
    if (accepted) accepted->assign(K, false);
    std::vector<std::vector<float>> rd(K), rf(K);
    int ns = 1, nt = 1;
    for (int k = 0; k < K; ++k) {
      if (!sources[k] || sources[k]->Empty() || !targets[k] || targets[k]->Empty()) { std::fprintf(stderr, "[ERROR] Empty cloud.\n"); return false; }
      rd[k] = DropNan(sources[k]->GetInnerCloud());
      rf[k] = DropNan(targets[k]->GetInnerCloud());
      if (rd[k].empty() || rf[k].empty()) return false;
      ns = std::max(ns, static_cast<int>(rd[k].size() / 3));
      nt = std::max(nt, static_cast<int>(rf[k].size() / 3));
    }
    if (!batch_arena_.Reserve(device_, std::max(2 * K, batch_arena_.slots), ns, nt)) return false;
    smhip_handle h = batch_arena_.handle;
    auto bad = [&]() { std::fprintf(stderr, "[ERROR] IcpPointMatcherHip::AlignBatch: %s\n", smhip_last_error(h)); return false; };
    for (int k = 0; k < K; ++k) {
      int m = 0;
      if (smhip_set_source_f32(h, K + k, rd[k].data(), 3, static_cast<int>(rd[k].size() / 3)) != SMHIP_OK ||
          smhip_set_target_f32(h, K + k, rf[k].data(), 3, nullptr, 0, static_cast<int>(rf[k].size() / 3)) != SMHIP_OK ||
          smhip_sample_source(h, K + k, k, prob_, static_cast<uint32_t>(seed_), &m) != SMHIP_OK ||
          smhip_prepare_target_from_target(h, K + k, k, &m) != SMHIP_OK) return bad();
    }
    smhip_icp_options o; smhip_icp_default_options(&o);
    o.max_iteration = 150; o.dist_outlier_ratio = 0.7f; o.early_exit = 1;
    o.nn_mode = nn_mode_; o.nn_epsilon = nn_epsilon_;
    if (smhip_icp_set_options(h, &o) != SMHIP_OK) return bad();
    std::vector<double> g(16 * static_cast<size_t>(K)), r(16 * static_cast<size_t>(K)), sc(K);
    for (int k = 0; k < K; ++k) std::memcpy(&g[16 * static_cast<size_t>(k)], guesses[k].data(), sizeof(double) * 16);
    if (smhip_icp_align_range(h, 0, K, g.data(), r.data(), sc.data(), nullptr) != SMHIP_OK) return bad();
    for (int k = 0; k < K; ++k) {
      double score = 0.0;
      if (smhip_icp_trimmed_score(h, K + k, &r[16 * static_cast<size_t>(k)], 0.7f, &score, nullptr) != SMHIP_OK) return bad();
      std::memcpy((*results)[k].data(), &r[16 * static_cast<size_t>(k)], sizeof(double) * 16);
      (*scores)[k] = score;
      if (accepted) (*accepted)[k] = score >= 0.6;
    }
    return true;
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
    std::fprintf(stderr, "[ERROR] IcpPointMatcherHip::Align: %s\n", arena_.handle ? smhip_last_error(arena_.handle) : "no handle");
    result = guess;
    return false;
  }
  bool EnsureHandle(int ns, int nt, bool* recreated = nullptr) {
    const bool first = arena_.handle == nullptr;
    const int init = max_points_ > 0 ? max_points_ : 1;
    // the raw reference is also the input of the device CalculateNormals, whose scratch is sized by max(ns, nt) capacity
    if (!arena_.Reserve(device_, 2, std::max(ns, first ? init : 0), std::max(nt, first ? init : 0), recreated)) {
      SMHIP_CHECK(!first || smhip_device_count() > 0, "no usable MI355X (gfx950) device: there is no CPU fallback");
      return false;
    }
    return true;
  }
  std::vector<float> reading_, reference_;     // xyz of the non-NaN points
  bool reading_on_device_ = false, reference_on_device_ = false;
  float prob_ = 0.9f;                           // icp_pointmatcher.cc:172
  int32_t seed_ = 0;
  int32_t nn_mode_ = SMHIP_NN_GRID;             // SMHIP_NN_NABO = libpointmatcher's KDTreeMatcher as configured (:186-191): libnabo, epsilon 3.16
  float nn_epsilon_ = 3.16f;
  int32_t device_ = 0;
  int32_t max_points_;
  DeviceArena arena_;
  DeviceArena batch_arena_;                    // AlignBatch's 2 K-slot handle
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
