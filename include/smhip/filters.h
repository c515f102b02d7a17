// smhip/filters.h -- C++ mirror of the reference's pre-filter plugin surface on top of the C ABI (include/smhip.h).
//
// Mirrors, name for name,
//   /root/reference/pre_processors/processor_interface.h:33-63   ProcesserInterface (SetInputCloud, Inliers, Outliers)
//   /root/reference/pre_processors/xml_interface.h:36-75         XmlInterface::SetValue by parameter name
//   /root/reference/pre_processors/filter_interface.h:38-64      filter::Interface (InitFromXmlText, ConfigsValid, Filter)
//   /root/reference/pre_processors/filter_{range,axis_range,bounding_box,random_sample,voxel_grid}.{h,cc}
//   /root/reference/pre_processors/filter_factory.{h,cc}         Factory: the <filters> chain
// free of glog / pugixml / pcl.  Every filter runs on the GPU through smhip_filter_chain_f32; a Factory runs its whole
// chain in one call and can hand the result to a matcher slot without a host round trip (FilterToSource).
// Header-only; link with -lsmhip.
#ifndef SMHIP_FILTERS_H_
#define SMHIP_FILTERS_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "smhip.h"
#include "smhip/registrator.h"

namespace smhip {
namespace data {
struct InnerCloudType {                     // cloud_types.h:60-77 (the stamp is carried through FilterPrepare)
  int64_t stamp = 0;
  std::vector<InnerPointType> points;
  using Ptr = std::shared_ptr<InnerCloudType>;
};
}  // namespace data

namespace pre_processers {
namespace filter {

// One device context shared by the filters of a process (capacity = largest cloud they will see).
class DeviceContext {
 public:
  explicit DeviceContext(int device = 0, int max_points = 1 << 18) {
    const smhip_status s = smhip_create(device, nullptr, 1, max_points, max_points, &handle_);
    if (s != SMHIP_OK) { std::fprintf(stderr, "[FATAL] no usable MI355X (gfx950) device: there is no CPU fallback\n"); std::abort(); }
  }
  ~DeviceContext() { if (handle_) smhip_destroy(handle_); }
  DeviceContext(const DeviceContext&) = delete;
  DeviceContext& operator=(const DeviceContext&) = delete;
  smhip_handle handle() const { return handle_; }
  static std::shared_ptr<DeviceContext>& Default() { static std::shared_ptr<DeviceContext> c; if (!c) c.reset(new DeviceContext()); return c; }

 private:
  smhip_handle handle_ = nullptr;
};

class Interface {
 public:
  Interface() = default;
  virtual ~Interface() = default;
  Interface(const Interface&) = delete;
  Interface& operator=(const Interface&) = delete;

  void SetDeviceContext(const std::shared_ptr<DeviceContext>& c) { context_ = c; }

  virtual void SetInputCloud(const data::InnerCloudType::Ptr& cloud) {          // processor_interface.h:42-51
    inliers_.clear(); outliers_.clear();
    if (!cloud || cloud->points.empty()) { std::fprintf(stderr, "[WARNING] cloud empty, do nothing!\n"); inner_cloud_ = nullptr; return; }
    inner_cloud_ = cloud;
  }
  virtual const std::vector<int>& Inliers() const { return inliers_; }
  virtual const std::vector<int>& Outliers() const { return outliers_; }

  // filter_interface.cc:30-74: <filter name="..."><param type="0|1" name="...">v</param>...</filter>
  bool InitFromXmlText(const char* xml_text) {
    const std::string t(xml_text ? xml_text : "");
    const size_t f = t.find("<filter");
    if (f == std::string::npos) { std::fprintf(stderr, "[FATAL] invalid xml text.\n"); std::abort(); }
    if (Attribute(t, f, "name") != GetName() || GetName().empty()) return false;
    size_t pos = f;
    while ((pos = t.find("<param", pos)) != std::string::npos) {
      const std::string type = Attribute(t, pos, "type"), name = Attribute(t, pos, "name");
      const size_t open = t.find('>', pos), close = t.find("</param>", pos);
      if (open == std::string::npos || close == std::string::npos) break;
      const std::string text = t.substr(open + 1, close - open - 1);
      bool all_right = false;
      if (type == "0") all_right = SetValue(name, static_cast<double>(std::atoi(text.c_str())));
      else if (type == "1") all_right = SetValue(name, std::atof(text.c_str()));
      if (!all_right) { std::fprintf(stderr, "[FATAL] Check failed: all_right (param %s)\n", name.c_str()); std::abort(); }   // :58
      pos = close;
    }
    return ConfigsValid();
  }
  virtual bool ConfigsValid() const { return smhip_filter_config_valid(&desc_) != 0; }
  virtual std::shared_ptr<Interface> CreateNewInstance() = 0;
  virtual std::string GetName() const = 0;

  // Filter and output the inlier points to cloud
  virtual void Filter(const data::InnerCloudType::Ptr& cloud) {
    if (!cloud || !inner_cloud_) { std::fprintf(stderr, "[WARNING] nullptr cloud, do nothing!\n"); return; }
    const std::vector<smhip_filter_desc> chain(1, desc_);
    RunChain(chain, cloud, true);
  }
  const smhip_filter_desc& Desc() const { return desc_; }

 protected:
  bool SetValue(const std::string& name, double value) {                       // xml_interface.h:46-66
    auto it = params_.find(name);
    if (it == params_.end()) return false;
    if (it->second == -1) desc_.axis_index = static_cast<int32_t>(value);
    else desc_.p[it->second] = static_cast<float>(value);
    return true;
  }
  void RunChain(const std::vector<smhip_filter_desc>& chain, const data::InnerCloudType::Ptr& cloud, bool indices) {
    if (!context_) context_ = DeviceContext::Default();
    smhip_handle h = context_->handle();
    cloud->points.clear();                                                     // FilterPrepare, filter_interface.cc:86-92
    cloud->stamp = inner_cloud_->stamp;
    inliers_.clear(); outliers_.clear();
    const int n = static_cast<int>(inner_cloud_->points.size());
    int m = 0;
    smhip_status s = smhip_filter_chain_f32(h, &inner_cloud_->points[0].x, 5, n, chain.data(), static_cast<int>(chain.size()), &m);
    if (s != SMHIP_OK) { std::fprintf(stderr, "[FATAL] filter chain: %s (%s)\n", smhip_status_string(s), smhip_last_error(h)); std::abort(); }
    cloud->points.resize(m);
    std::vector<int32_t> src(m);
    if (m > 0) s = smhip_filter_get_output(h, &cloud->points[0].x, src.data(), m);
    if (s != SMHIP_OK) { std::fprintf(stderr, "[FATAL] filter output: %s\n", smhip_last_error(h)); std::abort(); }
    if (indices && (m == 0 || src[0] >= 0)) {                                  // a VoxelGrid keeps no index lists (filter_voxel_grid.cc)
      inliers_.assign(src.begin(), src.end());
      outliers_.reserve(n - m);
      int k = 0;
      for (int i = 0; i < n; ++i) { if (k < m && src[k] == i) ++k; else outliers_.push_back(i); }
    }
  }
  static std::string Attribute(const std::string& t, size_t from, const char* key) {
    const size_t end = t.find('>', from);
    const std::string k = std::string(key) + "=\"";
    const size_t a = t.find(k, from);
    if (a == std::string::npos || a > end) return "";
    const size_t b = t.find('"', a + k.size());
    return t.substr(a + k.size(), b - a - k.size());
  }

  smhip_filter_desc desc_{};
  std::map<std::string, int> params_;        // parameter name -> p[] slot (-1 = axis_index)
  data::InnerCloudType::Ptr inner_cloud_;
  std::vector<int> inliers_, outliers_;
  std::shared_ptr<DeviceContext> context_;
};

#define SMHIP_FILTER_CLASS(Name, Type, ...)                                                            \
  class Name : public Interface {                                                                      \
   public:                                                                                             \
    Name() { smhip_filter_default(Type, &desc_); params_ = std::map<std::string, int> __VA_ARGS__; }   \
    std::shared_ptr<Interface> CreateNewInstance() override { return std::make_shared<Name>(); }       \
    std::string GetName() const override { return #Name; }                                             \
  }

SMHIP_FILTER_CLASS(Range, SMHIP_FILTER_RANGE, {{"min_range", 0}, {"max_range", 1}});                                  // filter_range.cc:33-38
SMHIP_FILTER_CLASS(AxisRange, SMHIP_FILTER_AXIS_RANGE, {{"min", 0}, {"max", 1}, {"axis_index", -1}});                   // filter_axis_range.cc:28-32
SMHIP_FILTER_CLASS(VoxelGrid, SMHIP_FILTER_VOXEL_GRID, {{"voxel_size", 0}});                                            // filter_voxel_grid.cc:28-30
SMHIP_FILTER_CLASS(BoundingBoxRemoval, SMHIP_FILTER_BOUNDING_BOX_REMOVAL,
                   {{"min_x", 0}, {"min_y", 1}, {"min_z", 2}, {"max_x", 3}, {"max_y", 4}, {"max_z", 5}});              // filter_bounding_box.cc:31-38

// filter_random_sample.cc:28-30; `seed` selects the random stream (the reference reseeds from std::random_device per call)
class RandomSampler : public Interface {
 public:
  RandomSampler() { smhip_filter_default(SMHIP_FILTER_RANDOM_SAMPLER, &desc_); params_ = {{"sampling_rate", 0}}; }
  std::shared_ptr<Interface> CreateNewInstance() override { return std::make_shared<RandomSampler>(); }
  std::string GetName() const override { return "RandomSampler"; }
  void SetSeed(uint32_t seed) { desc_.seed = seed; }
  void Filter(const data::InnerCloudType::Ptr& cloud) override { Interface::Filter(cloud); desc_.seed += 0x9e3779b9u; }   // a fresh stream per call
};

// filter_factory.cc:47-106
class Factory : public Interface {
 public:
  Factory() {
    supported_filters_.emplace("RandomSampler", std::make_shared<RandomSampler>());
    supported_filters_.emplace("Range", std::make_shared<Range>());
    supported_filters_.emplace("VoxelGrid", std::make_shared<VoxelGrid>());
    supported_filters_.emplace("AxisRange", std::make_shared<AxisRange>());
    supported_filters_.emplace("BoundingBoxRemoval", std::make_shared<BoundingBoxRemoval>());
  }
  std::shared_ptr<Interface> CreateNewInstance() override { return std::make_shared<Factory>(); }
  std::string GetName() const override { return ""; }
  bool ConfigsValid() const override { return true; }

  void InitFromXmlText(const char* text) {                                     // <filters> <filter .../> ... </filters>
    const std::string t(text ? text : "");
    size_t pos = 0;
    while ((pos = t.find("<filter ", pos)) != std::string::npos) {
      const std::string name = Attribute(t, pos, "name");
      const size_t tag_end = t.find('>', pos);
      const bool self_closing = tag_end != std::string::npos && tag_end > 0 && t[tag_end - 1] == '/';
      const size_t close = self_closing ? tag_end : t.find("</filter>", pos);
      if (close == std::string::npos) break;
      const size_t end = self_closing ? tag_end + 1 : close + 9;
      auto it = supported_filters_.find(name);
      if (it == supported_filters_.end()) {
        std::fprintf(stderr, "[XML] %s not supported yet.\n", name.c_str());    // filter_factory.cc:55-58
      } else {
        auto f = it->second->CreateNewInstance();
        f->InitFromXmlText(t.substr(pos, end - pos).c_str());
        filters_.push_back(f);
      }
      pos = end;
    }
  }
  void Filter(const data::InnerCloudType::Ptr& cloud) override {              // filter_factory.cc:83-106
    if (!cloud || !inner_cloud_) { std::fprintf(stderr, "[WARNING] nullptr cloud, do nothing!\n"); return; }
    RunChain(Chain(), cloud, false);
  }
  // the same chain, the result handed to a matcher's source slot on the device (no host copy of the filtered cloud)
  bool FilterToSource(smhip_handle matcher, int slot, int* n_out) {
    if (!inner_cloud_) return false;
    const std::vector<smhip_filter_desc> chain = Chain();
    int m = 0;
    if (smhip_filter_chain_f32(matcher, &inner_cloud_->points[0].x, 5, static_cast<int>(inner_cloud_->points.size()), chain.data(),
                               static_cast<int>(chain.size()), &m) != SMHIP_OK) return false;
    if (n_out) *n_out = m;
    return smhip_filter_output_to_source(matcher, slot) == SMHIP_OK;
  }
  size_t size() const { return filters_.size(); }

 private:
  std::vector<smhip_filter_desc> Chain() const {
    std::vector<smhip_filter_desc> c;
    for (const auto& f : filters_) c.push_back(f->Desc());
    return c;
  }
  std::vector<std::shared_ptr<Interface>> filters_;
  std::map<std::string, std::shared_ptr<Interface>> supported_filters_;
};

}  // namespace filter
}  // namespace pre_processers
}  // namespace smhip

#endif  // SMHIP_FILTERS_H_
