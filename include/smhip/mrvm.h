// smhip/mrvm.h -- C++ mirror of static_map::MultiResolutionVoxelMap (/root/reference/builder/multi_resolution_voxel_map.h:67-131)
// over the device implementation in libsmhip.so: same settings struct, same three calls the map builder makes
// (builder/map_builder.cc:832-900): Initialise, InsertPointCloud(cloud, origin), OutputToPointCloud(threshold, cloud).
// Header-only; link with -lsmhip.
#ifndef SMHIP_MRVM_H_
#define SMHIP_MRVM_H_

#include <cstdio>
#include <memory>
#include <vector>

#include "../smhip.h"
#include "registrator.h"

namespace smhip {

struct MrvmSettings {                      // multi_resolution_voxel_map.h:54-65
  bool output_average = false;
  bool output_rgb = false;
  bool use_max_intensity = true;
  float prob_threshold = 0.6f;
  float low_resolution = 1.f;              // "not in use any more"
  float high_resolution = 0.1f;
  float hit_prob = 0.55f;
  float miss_prob = 0.48f;
  float z_offset = 0.f;
  int max_point_num_in_cell = 10;
};

struct PointXYZI { float x, y, z, intensity; };   // pcl::PointXYZI's payload
struct PointXYZRGB {                                // pcl::PointXYZRGB's payload: x y z + the packed colour in a float's bits
  float x, y, z;
  union { float rgb; struct { unsigned char b, g, r, a; }; };
};

class MultiResolutionVoxelMapHip {
 public:
  using InnerCloud = std::vector<data::InnerPointType>;        // data::InnerCloudType::points
  // table_log2 / max_cloud_points: the voxel table's first size (2^table_log2 slots; it doubles between inserts as the map grows, like
  // the reference's std::map) and the largest cloud of one insert
  explicit MultiResolutionVoxelMapHip(int device = 0, int table_log2 = 24, int max_cloud_points = 1 << 18)
      : device_(device), table_log2_(table_log2), max_cloud_points_(max_cloud_points) {}
  ~MultiResolutionVoxelMapHip() { if (handle_) smhip_mrvm_destroy(handle_); }
  MultiResolutionVoxelMapHip(const MultiResolutionVoxelMapHip&) = delete;
  MultiResolutionVoxelMapHip& operator=(const MultiResolutionVoxelMapHip&) = delete;

  void Initialise(const MrvmSettings& settings) {              // .cc:46-53
    SMHIP_CHECK(settings.max_point_num_in_cell > 0, "CHECK_GT(settings_.max_point_num_in_cell, 0)");
    settings_ = settings;
    if (handle_) { smhip_mrvm_destroy(handle_); handle_ = nullptr; }
    smhip_mrvm_settings s;
    smhip_mrvm_default_settings(&s);
    s.prob_threshold = settings.prob_threshold; s.high_resolution = settings.high_resolution; s.hit_prob = settings.hit_prob;
    s.miss_prob = settings.miss_prob; s.z_offset = settings.z_offset; s.max_point_num_in_cell = settings.max_point_num_in_cell;
    s.use_max_intensity = settings.use_max_intensity ? 1 : 0;
    const smhip_status st = smhip_mrvm_create(device_, table_log2_, max_cloud_points_, &s, &handle_);
    SMHIP_CHECK(st == SMHIP_OK, "smhip_mrvm_create failed (no gfx950 device? there is no CPU fallback)");
  }
  void SetOffsetZ(const float& offset) { settings_.z_offset = offset; if (handle_) smhip_mrvm_set_offset_z(handle_, offset); }   // .cc:55-57

  // .cc:59-131; origin = frame->GlobalTranslation() (map_builder.cc:847-848).  false: the device refused the cloud (printed).
  bool InsertPointCloud(const InnerCloud& cloud, const float origin[3]) {
    if (cloud.empty()) { std::fprintf(stderr, "[ERROR] cloud is empty.\n"); return false; }   // PRINT_ERROR + return, :61-64
    SMHIP_CHECK(handle_ != nullptr, "InsertPointCloud before Initialise");
    static_assert(sizeof(data::InnerPointType) == 5 * sizeof(float), "InnerPointType is five floats");
    const smhip_status st = smhip_mrvm_insert_f32(handle_, &cloud[0].x, 5, static_cast<int>(cloud.size()), origin);
    if (st != SMHIP_OK) { std::fprintf(stderr, "[ERROR] MultiResolutionVoxelMapHip::InsertPointCloud: %s\n", smhip_mrvm_last_error(handle_)); return false; }
    // applied, with something to say (points beyond the coordinate range skipped, table filling up): the reference has neither limit
    if (smhip_mrvm_last_error(handle_)[0]) std::fprintf(stderr, "[WARNING] MultiResolutionVoxelMapHip::InsertPointCloud: %s\n", smhip_mrvm_last_error(handle_));
    return true;
  }
  // .cc:125-170 (PointXYZI; one averaged point per voxel with settings_.output_average)
  void OutputToPointCloud(const float threshold, std::vector<PointXYZI>* cloud) {
    static_assert(sizeof(PointXYZI) == 4 * sizeof(float), "PointXYZI is four floats");
    Output(threshold, settings_.output_average ? SMHIP_MRVM_AVERAGE : 0, cloud);
  }
  // .cc:172-216 (PointXYZRGB: grey = min(255, max_intensity * 1.4))
  void OutputToPointCloud(const float threshold, std::vector<PointXYZRGB>* cloud) {
    static_assert(sizeof(PointXYZRGB) == 4 * sizeof(float), "PointXYZRGB is four floats");
    Output(threshold, SMHIP_MRVM_RGB | (settings_.output_average ? SMHIP_MRVM_AVERAGE : 0), cloud);
  }
  int VoxelCount() const { int n = 0; if (handle_) smhip_mrvm_voxel_count(handle_, &n); return n; }

 private:
  template <typename P>
  void Output(const float threshold, int flags, std::vector<P>* cloud) {
    SMHIP_CHECK(cloud != nullptr && handle_ != nullptr, "OutputToPointCloud: null cloud / not initialised");
    cloud->clear();
    int n = 0;
    if (smhip_mrvm_output_ex(handle_, threshold, flags, nullptr, 0, &n) != SMHIP_OK || n <= 0) return;
    cloud->resize(static_cast<size_t>(n));
    int m = 0;
    if (smhip_mrvm_output_ex(handle_, threshold, flags, &(*cloud)[0].x, n, &m) != SMHIP_OK) cloud->clear();
    else cloud->resize(static_cast<size_t>(m < n ? m : n));
  }
  int device_, table_log2_, max_cloud_points_;
  MrvmSettings settings_;
  smhip_mrvm_handle handle_ = nullptr;
};

}  // namespace smhip
#endif  // SMHIP_MRVM_H_
