// Replays /root/reference/pre_processors/test/test_filter_{range,axis_range,bounding_box,random_sample,voxel_grid}.cc
// against the C++ mirror in include/smhip/filters.h (Boost.Test is not available here: plain checks, JSON verdict).
#include <cmath>
#include <cstdio>
#include <random>

#include "smhip/filters.h"

using namespace smhip::pre_processers::filter;
using smhip::data::InnerCloudType;
using smhip::data::InnerPointType;

static int g_fail = 0;
#define CHECK_T(c) do { if (!(c)) { ++g_fail; std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

static InnerCloudType::Ptr CreateRandomInnerCloud(int size, unsigned seed) {       // test/test_helper.cc:49-61 with a fixed seed
  std::mt19937 gen(seed);
  std::uniform_real_distribution<> distrib(0., 100.);
  InnerCloudType::Ptr cloud(new InnerCloudType);
  cloud->stamp = 1000000000;
  for (int i = 0; i < size; ++i) {
    InnerPointType p;
    p.x = distrib(gen); p.y = distrib(gen); p.z = distrib(gen); p.intensity = distrib(gen);
    cloud->points.push_back(p);
  }
  return cloud;
}

int main() {
  {  // test_filter_range.cc
    Range f;
    auto raw = CreateRandomInnerCloud(1000, 1);
    f.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    f.Filter(out);
    CHECK_T(out->stamp == raw->stamp);
    CHECK_T(out->points.size() == raw->points.size());
    for (size_t i = 0; i < out->points.size(); ++i) CHECK_T(out->points[i].x == raw->points[i].x && out->points[i].intensity == raw->points[i].intensity);
    CHECK_T(f.InitFromXmlText("<filter name=\"Range\" ><param type=\"1\" name=\"min_range\"> 20. </param><param type=\"1\" name=\"max_range\"> 80. </param></filter>"));
    f.Filter(out);
    CHECK_T(!out->points.empty());
    for (const auto& p : out->points) { const float r = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); CHECK_T(r <= 80. && r >= 20.); }
    CHECK_T(f.Inliers().size() == out->points.size() && f.Inliers().size() + f.Outliers().size() == raw->points.size());
  }
  {  // test_filter_axis_range.cc
    AxisRange f;
    CHECK_T(!f.InitFromXmlText("<filter name=\"AxisRangeXXX\" />"));
    CHECK_T(!f.InitFromXmlText("<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> 90. </param><param type=\"1\" name=\"max\"> 80. </param></filter>"));
    AxisRange g;
    CHECK_T(!g.InitFromXmlText("<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> 10. </param><param type=\"1\" name=\"max\"> 80. </param><param type=\"0\" name=\"axis_index\"> -1 </param></filter>"));
    AxisRange a;
    auto raw = CreateRandomInnerCloud(1000, 2);
    a.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    a.Filter(out);
    CHECK_T(out->points.size() == raw->points.size());
    CHECK_T(a.InitFromXmlText("<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> 50. </param></filter>"));
    a.Filter(out);
    for (const auto& p : out->points) CHECK_T(p.z >= 50.);
    CHECK_T(a.InitFromXmlText("<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> 60. </param><param type=\"1\" name=\"max\"> 80. </param><param type=\"0\" name=\"axis_index\"> 1 </param></filter>"));
    a.Filter(out);
    for (const auto& p : out->points) CHECK_T(p.y >= 60. && p.y <= 80.);
    CHECK_T(a.InitFromXmlText("<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> 10. </param><param type=\"1\" name=\"max\"> 70. </param><param type=\"0\" name=\"axis_index\"> 0 </param></filter>"));
    a.Filter(out);
    for (const auto& p : out->points) CHECK_T(p.x >= 10. && p.x <= 70.);
  }
  {  // test_filter_bounding_box.cc
    BoundingBoxRemoval f;
    CHECK_T(!f.InitFromXmlText("<filter name=\"BoundingBoxRemoval\" ><param type=\"1\" name=\"min_x\"> 90. </param><param type=\"1\" name=\"max_x\"> 80. </param></filter>"));
    BoundingBoxRemoval b;
    auto raw = CreateRandomInnerCloud(1000, 3);
    b.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    b.Filter(out);
    CHECK_T(out->points.size() == 0 && out->stamp == raw->stamp);
    CHECK_T(b.InitFromXmlText("<filter name=\"BoundingBoxRemoval\" ><param type=\"1\" name=\"min_x\"> 10. </param><param type=\"1\" name=\"max_x\"> 80. </param>"
                              "<param type=\"1\" name=\"min_y\"> 20. </param><param type=\"1\" name=\"max_y\"> 70. </param>"
                              "<param type=\"1\" name=\"min_z\"> 30. </param><param type=\"1\" name=\"max_z\"> 80. </param></filter>"));
    b.Filter(out);
    CHECK_T(!out->points.empty());
    for (const auto& p : out->points) CHECK_T(!(p.x >= 10. && p.x <= 80. && p.y >= 20. && p.y <= 70. && p.z >= 30. && p.z <= 80.));
  }
  {  // test_filter_random_sample.cc
    RandomSampler f;
    CHECK_T(f.InitFromXmlText("<filter name=\"RandomSampler\" />"));
    CHECK_T(!f.InitFromXmlText("<filter name=\"RandomSamplerSSS\" />"));
    RandomSampler bad;
    CHECK_T(!bad.InitFromXmlText("<filter name=\"RandomSampler\" ><param type=\"1\" name=\"sampling_rate\"> 1.5 </param></filter>"));
    CHECK_T(f.InitFromXmlText("<filter name=\"RandomSampler\" ><param type=\"1\" name=\"sampling_rate\"> 0.5 </param></filter>"));
    const int n = 100000;
    auto raw = CreateRandomInnerCloud(n, 4);
    f.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    size_t last = 0; bool varies = false;
    for (int i = 0; i < 20; ++i) {
      f.Filter(out);
      const double frac = static_cast<double>(out->points.size()) / n;
      CHECK_T(frac > 0.48 && frac < 0.52);
      if (i && out->points.size() != last) varies = true;
      last = out->points.size();
    }
    CHECK_T(varies);
  }
  {  // test_filter_voxel_grid.cc
    VoxelGrid f;
    CHECK_T(!f.InitFromXmlText("<filter name=\"VoxelGrid\" ><param type=\"1\" name=\"voxel_size\"> 0. </param></filter>"));
    InnerCloudType::Ptr raw(new InnerCloudType);
    raw->stamp = 42;
    for (int x = 0; x < 10; ++x)
      for (int y = 0; y < 10; ++y) { InnerPointType p; p.x = x * 0.1f + 0.02f; p.y = y * 0.1f + 0.02f; p.z = 0.1f; raw->points.push_back(p); }
    VoxelGrid v;
    v.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    const char* cfg[3] = {"0.1", "0.2", "0.4"};
    const size_t want[3] = {100, 36, 9};
    for (int k = 0; k < 3; ++k) {
      const std::string t = std::string("<filter name=\"VoxelGrid\" ><param type=\"1\" name=\"voxel_size\"> ") + cfg[k] + " </param></filter>";
      CHECK_T(v.InitFromXmlText(t.c_str()));
      v.Filter(out);
      CHECK_T(out->stamp == raw->stamp);
      CHECK_T(out->points.size() == want[k]);
    }
  }
  {  // Factory with the chain of config/lidar_only_kitti.xml:18-41
    Factory fac;
    fac.InitFromXmlText("<filters><filter name=\"Range\" ><param type=\"1\" name=\"min_range\"> 5. </param></filter>"
                        "<filter name=\"AxisRange\" ><param type=\"1\" name=\"min\"> -2. </param></filter>"
                        "<filter name=\"GroundRemoval2\" ><param type=\"1\" name=\"r_min\"> 0.1 </param></filter>"
                        "<filter name=\"RandomSampler\" ><param type=\"1\" name=\"sampling_rate\"> 0.5 </param></filter></filters>");
    CHECK_T(fac.size() == 3);
    auto raw = CreateRandomInnerCloud(20000, 5);
    for (auto& p : raw->points) { p.x -= 50.f; p.y -= 50.f; p.z -= 50.f; }
    fac.SetInputCloud(raw);
    InnerCloudType::Ptr out(new InnerCloudType);
    fac.Filter(out);
    CHECK_T(!out->points.empty() && out->points.size() < raw->points.size());
    for (const auto& p : out->points) { CHECK_T(std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z) >= 5.f); CHECK_T(p.z >= -2.f); }
    int m = 0;
    CHECK_T(fac.FilterToSource(DeviceContext::Default()->handle(), 0, &m) && m > 0);
  }
  std::printf("{\"failed\": %d}\n", g_fail);
  return g_fail ? 1 : 0;
}
