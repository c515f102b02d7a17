// The C++ mirror of MultiResolutionVoxelMap (include/smhip/mrvm.h) driven the way builder/map_builder.cc:832-900 drives
// the reference: Initialise(settings), InsertPointCloud(frame cloud in the map frame, frame translation) per frame,
// OutputToPointCloud(prob_threshold).  usage: test_mrvm cloud0.bin ox oy oz cloud1.bin ox oy oz ... (rows of 5 floats)
#define SMHIP_REGISTRATOR_THROW_ON_CHECK 1
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "smhip/mrvm.h"

int main(int argc, char** argv) {
  smhip::MultiResolutionVoxelMapHip map(0, 20, 1 << 17), avg_map(0, 20, 1 << 17);
  smhip::MrvmSettings s;
  map.Initialise(s);
  smhip::MrvmSettings sa = s;
  sa.output_average = true;                          // the same map, read out as one mean point per voxel (.cc:136-151)
  avg_map.Initialise(sa);
  bool empty_refused = !map.InsertPointCloud({}, (const float[3]){0, 0, 0});
  for (int a = 1; a + 3 < argc; a += 4) {
    std::ifstream f(argv[a], std::ios::binary | std::ios::ate);
    if (!f) return 2;
    const size_t bytes = static_cast<size_t>(f.tellg());
    std::vector<smhip::data::InnerPointType> cloud(bytes / sizeof(smhip::data::InnerPointType));
    f.seekg(0); f.read(reinterpret_cast<char*>(cloud.data()), static_cast<std::streamsize>(cloud.size() * sizeof(smhip::data::InnerPointType)));
    const float origin[3] = {(float)std::atof(argv[a + 1]), (float)std::atof(argv[a + 2]), (float)std::atof(argv[a + 3])};
    if (!map.InsertPointCloud(cloud, origin)) return 3;
    if (!avg_map.InsertPointCloud(cloud, origin)) return 3;
  }
  std::vector<smhip::PointXYZI> out;
  map.OutputToPointCloud(s.prob_threshold, &out);
  double sum = 0;
  for (const auto& p : out) sum += (double)p.x + 2.0 * p.y + 3.0 * p.z + 0.001 * p.intensity;
  std::vector<smhip::PointXYZRGB> rgb;
  map.OutputToPointCloud(s.prob_threshold, &rgb);                    // the PointXYZRGB overload (.cc:172-216)
  double rgb_sum = 0;
  bool grey = true;
  for (const auto& p : rgb) { rgb_sum += (double)p.x + 2.0 * p.y + 3.0 * p.z + 0.001 * p.r; grey = grey && p.r == p.g && p.g == p.b; }
  std::vector<smhip::PointXYZI> avg;
  avg_map.OutputToPointCloud(sa.prob_threshold, &avg);
  double avg_sum = 0;
  for (const auto& p : avg) avg_sum += (double)p.x + 2.0 * p.y + 3.0 * p.z + 0.001 * p.intensity;
  std::printf("{\"voxels\": %d, \"output_points\": %zu, \"checksum\": %.9g, \"empty_refused\": %s, \"rgb_points\": %zu, \"rgb_checksum\": %.9g, "
              "\"grey\": %s, \"average_points\": %zu, \"average_checksum\": %.9g}\n",
              map.VoxelCount(), out.size(), sum, empty_refused ? "true" : "false", rgb.size(), rgb_sum, grey ? "true" : "false", avg.size(), avg_sum);
  return 0;
}
