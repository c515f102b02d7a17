// MapBuilder::ScanMatchProcessing (include/smhip/front_end.h) over a short synthetic drive.
// argv: n_scans dir [vx vy [device_target_prep]]  (dir holds 0000000000.bin ... KITTI rows)  -> JSON with the pose of
// every scan and the wall time ProcessCloud took for it
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "smhip/front_end.h"

namespace reg = smhip::registrator;
using smhip::data::InnerPointCloudData;
using smhip::data::InnerPointType;

static std::vector<InnerPointType> ReadKittiBin(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<InnerPointType> pts;
  float row[4];
  while (f.read(reinterpret_cast<char*>(row), sizeof(row))) { InnerPointType p; p.x = row[0]; p.y = row[1]; p.z = row[2]; p.intensity = row[3]; pts.push_back(p); }
  for (size_t i = 0; i < pts.size(); ++i) pts[i].factor = static_cast<float>(static_cast<double>(i) / pts.size());
  return pts;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int n = std::atoi(argv[1]);
  reg::MatcherOptions opt;
  opt.type = reg::kFastIcp;                                        // config/lidar_only_kitti.xml:49
  opt.registrator_options_node = "<param name=\"max_iteration\"> 100 </param><param name=\"dist_outlier_ratio\"> 0.7 </param>";
  auto matcher = reg::CreateMatcher(opt);
  if (!matcher) return 3;
  smhip::front_end::MotionFilter mf;
  mf.translation_range = 0.5f;                                     // config/lidar_only_kitti.xml:76
  const bool device_prep = argc > 5 && std::atoi(argv[5]) != 0;
  smhip::front_end::ScanMatcherFrontEnd fe(matcher, mf, true, device_prep);
  if (device_prep && !fe.DeviceTargetPrep()) return 4;
  if (argc > 3) fe.Extrapolator().InitRoughLinearVelocity(std::atof(argv[3]), argc > 4 ? std::atof(argv[4]) : 0.0, 0.0);   // pose_extrapolator.cc:210-214
  std::printf("{\"frames\": [");
  for (int k = 0; k < n; ++k) {
    char name[64];
    std::snprintf(name, sizeof(name), "/%010d.bin", k);
    InnerPointCloudData::Ptr cloud(new InnerPointCloudData(ReadKittiBin(std::string(argv[2]) + name)));
    const auto t0 = std::chrono::steady_clock::now();
    const auto r = fe.ProcessCloud(cloud, 0.1 * k);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%s{\"key\": %s, \"matched\": %s, \"score\": %.9g, \"ms\": %.3f, \"pose\": [", k ? ", " : "", r.new_key_frame ? "true" : "false",
                r.matched ? "true" : "false", r.score, ms);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) std::printf("%.17g%s", r.pose(i, j), (i == 3 && j == 3) ? "" : ", ");
    std::printf("]}");
  }
  std::printf("]}\n");
  return 0;
}
