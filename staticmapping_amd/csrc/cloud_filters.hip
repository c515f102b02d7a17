// cloud_filters.hip -- the front end's point-cloud pre-filters on the device.
//
// Reference: /root/reference/pre_processors/
//   filter_range.cc:46-91          Range               keep min_range <= |p| <= max_range (float norm)
//   filter_axis_range.cc:45-103    AxisRange           keep min <= p[axis] <= max
//   filter_bounding_box.cc:53-83   BoundingBoxRemoval  DROP the points inside [min, max] (common/bounding_box.cc:117-121)
//   filter_random_sample.cc:41-85  RandomSampler       keep a point when a uniform draw <= sampling_rate
//   filter_voxel_grid.cc:38-80     VoxelGrid           one average point (double sums) per lround(p / size) voxel
//   filter_factory.cc:83-106       Factory::Filter     the filters of <filters> applied in order
// They run on every scan right before the registrator (builder/data/data_collector.h, config/lidar_only_kitti.xml:18-41).
// All but VoxelGrid are order-preserving compactions: flag -> exclusive scan (rocPRIM building block) -> scatter;
// consecutive predicate filters share one pass.  VoxelGrid is a stable radix sort on the packed voxel index and one
// thread per voxel summing its points in arrival order in double, so each output point carries the reference's bits
// (the reference emits voxels in unordered_map order, i.e. unspecified; here they come out sorted by voxel index).
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>

#include "../../include/smhip.h"
#include "cloud_filters.h"

namespace smhip {

namespace {

constexpr int kMaxFused = 8;
struct Pred { int32_t type; int32_t axis; float p[6]; };
struct PredGroup { int32_t n; Pred f[kMaxFused]; };

// filter_range.cc:60-66 without fused multiply-adds (the reference's float expression, operation by operation)
__device__ __forceinline__ bool keep_point(const Pred& f, const float4 p) {
  switch (f.type) {
    case SMHIP_FILTER_RANGE: {
      const float r = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(p.x, p.x), __fmul_rn(p.y, p.y)), __fmul_rn(p.z, p.z)));
      return r >= f.p[0] && r <= f.p[1];
    }
    case SMHIP_FILTER_AXIS_RANGE: {
      const float v = f.axis == 0 ? p.x : (f.axis == 1 ? p.y : p.z);
      return !(v < f.p[0] || v > f.p[1]);                                  // filter_axis_range.cc:68-86
    }
    case SMHIP_FILTER_BOUNDING_BOX_REMOVAL: {
      const double x = p.x, y = p.y, z = p.z;                              // Eigen::Vector3d(point.x, ...), :72
      const bool inside = (x >= (double)f.p[0] && x <= (double)f.p[3]) && (y >= (double)f.p[1] && y <= (double)f.p[4]) &&
                          (z >= (double)f.p[2] && z <= (double)f.p[5]);
      return !inside;
    }
  }
  return true;
}

__global__ void filt_flags(const float4* pts, int n, PredGroup g, int32_t* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  bool k = true;
  for (int q = 0; q < g.n; ++q) k = k && keep_point(g.f[q], p);
  flag[i] = k ? 1 : 0;
}

// counter-based uniform in [0, 1): the reference seeds a mt19937 from std::random_device on every call
// (filter_random_sample.cc:57-59), so no particular stream can be reproduced; what is kept is the law:
// one uniform double per input point of THIS filter, kept when u <= sampling_rate.
__device__ __forceinline__ double sampler_uniform(uint32_t seed, uint32_t i) {
  unsigned long long z = ((unsigned long long)seed << 32 | i) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__global__ void filt_sample_flags(int n, uint32_t seed, float rate, int32_t* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = sampler_uniform(seed, (uint32_t)i) <= (double)rate ? 1 : 0;   // distr(eng) <= sampling_rate_, :68
}

__global__ void filt_scatter(const float4* pts, const float* fac, const int32_t* src, int n, const int32_t* flag, const int32_t* pos,
                             float4* out_pts, float* out_fac, int32_t* out_src, int32_t* count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) {
    const int o = pos[i];
    out_pts[o] = pts[i]; out_fac[o] = fac[i]; out_src[o] = src[i];
  }
  if (i == n - 1) count[0] = pos[i] + flag[i];
}

// ---- VoxelGrid ---------------------------------------------------------------------------------
constexpr long kVoxBias = 1l << 20;            // 21 bits per axis
__device__ __forceinline__ bool voxel_key(const float4 p, float size, unsigned long long& key) {
  const long ix = lroundf(p.x / size), iy = lroundf(p.y / size), iz = lroundf(p.z / size);   // filter_voxel_grid.cc:51-53
  const long a = ix + kVoxBias, b = iy + kVoxBias, c = iz + kVoxBias;
  const bool ok = a >= 0 && a < 2 * kVoxBias && b >= 0 && b < 2 * kVoxBias && c >= 0 && c < 2 * kVoxBias;
  key = ok ? (((unsigned long long)a << 42) | ((unsigned long long)b << 21) | (unsigned long long)c) : ~0ull;
  return ok;
}
__global__ void filt_voxel_keys(const float4* pts, int n, float size, unsigned long long* keys, int32_t* idx, int32_t* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long k;
  if (!voxel_key(pts[i], size, k)) atomicAdd(bad, 1);
  keys[i] = k; idx[i] = i;
}
__global__ void filt_voxel_heads(const unsigned long long* keys, int n, int32_t* head) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  head[s] = (s == 0 || keys[s] != keys[s - 1]) ? 1 : 0;
}
__global__ void filt_voxel_starts(const int32_t* head, const int32_t* incl, int n, int32_t* start, int32_t* count) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (head[s]) start[incl[s] - 1] = s;
  if (s == n - 1) count[0] = incl[s];
}
__global__ void filt_voxel_average(const float4* pts, const int32_t* idx, const int32_t* start, const int32_t* count, int n,
                                   float4* out_pts, float* out_fac, int32_t* out_src) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int V = count[0];
  if (v >= V) return;
  const int a = start[v], b = (v + 1 < V) ? start[v + 1] : n;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int s = a; s < b; ++s) { const float4 p = pts[idx[s]]; s0 += p.x; s1 += p.y; s2 += p.z; s3 += p.w; }   // :61-66
  const int size = b - a;
  out_pts[v] = make_float4((float)(s0 / size), (float)(s1 / size), (float)(s2 / size), (float)(s3 / size));   // :68-74
  out_fac[v] = 0.f;                              // InnerPointType result: factor keeps its default
  out_src[v] = -1;
}

__global__ void filt_init(const float4* in, int n, int stride5, const float* fac_in, float4* pts, float* fac, int32_t* src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pts[i] = in[i];
  // stride 5 rows carry their factor; KITTI rows get the collector's i / size (data_collector.h:202-204)
  fac[i] = stride5 ? fac_in[i] : (float)((double)i / (double)n);
  src[i] = i;
}

}  // namespace

struct FilterWorkspace {
  int cap = 0;
  float4* pts[2] = {nullptr, nullptr};
  float* fac[2] = {nullptr, nullptr};
  int32_t* src[2] = {nullptr, nullptr};
  int32_t *flag = nullptr, *pos = nullptr, *idx[2] = {nullptr, nullptr}, *start = nullptr, *counts = nullptr;
  unsigned long long* keys[2] = {nullptr, nullptr};
  float* fac_in = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int32_t* host_pinned = nullptr;
  int cur = 0;              // which of pts[2] holds the current cloud
  int n = 0;
  bool has_index = true;    // false once a VoxelGrid has run
};

FilterWorkspace* filt_create(int max_points) {
  FilterWorkspace* w = new FilterWorkspace();
  w->cap = max_points;
  const size_t N = (size_t)max_points;
  bool ok = true;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; };
  for (int k = 0; k < 2; ++k) {
    A((void**)&w->pts[k], N * sizeof(float4)); A((void**)&w->fac[k], N * 4); A((void**)&w->src[k], N * 4);
    A((void**)&w->idx[k], N * 4); A((void**)&w->keys[k], N * 8);
  }
  A((void**)&w->flag, N * 4); A((void**)&w->pos, N * 4); A((void**)&w->start, N * 4); A((void**)&w->counts, 16 * 4);
  A((void**)&w->fac_in, N * 4);
  if (ok) {
    size_t b1 = 0, b2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b1, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned)max_points, 0, 64, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b2, (const int32_t*)nullptr, (int32_t*)nullptr, 0, N, rocprim::plus<int32_t>(), (hipStream_t)0);
    w->tmp_bytes = std::max(b1, b2) + 256;
    A(&w->tmp, w->tmp_bytes);
  }
  if (ok && hipHostMalloc((void**)&w->host_pinned, 64) != hipSuccess) ok = false;
  if (!ok) { filt_destroy(w); return nullptr; }
  return w;
}

void filt_destroy(FilterWorkspace* w) {
  if (!w) return;
  for (int k = 0; k < 2; ++k) {
    (void)hipFree(w->pts[k]); (void)hipFree(w->fac[k]); (void)hipFree(w->src[k]); (void)hipFree(w->idx[k]); (void)hipFree(w->keys[k]);
  }
  (void)hipFree(w->flag); (void)hipFree(w->pos); (void)hipFree(w->start); (void)hipFree(w->counts); (void)hipFree(w->fac_in);
  (void)hipFree(w->tmp);
  if (w->host_pinned) (void)hipHostFree(w->host_pinned);
  delete w;
}

#define FCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

hipError_t filt_set_input(FilterWorkspace* w, hipStream_t st, const float4* staged_dev, const float* factor_host_or_null, int n) {
  if (!w || n < 0 || n > w->cap) return hipErrorInvalidValue;
  w->cur = 0; w->n = n; w->has_index = true;
  if (n == 0) return hipSuccess;
  if (factor_host_or_null) FCHK(hipMemcpyAsync(w->fac_in, factor_host_or_null, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(filt_init, dim3((n + 255) / 256), dim3(256), 0, st, staged_dev, n, factor_host_or_null ? 1 : 0, w->fac_in,
                     w->pts[0], w->fac[0], w->src[0]);
  return hipGetLastError();
}

static hipError_t compact(FilterWorkspace* w, hipStream_t st) {      // flags are in w->flag
  const int n = w->n, c = w->cur, o = 1 - c;
  size_t bytes = w->tmp_bytes;
  FCHK(rocprim::exclusive_scan(w->tmp, bytes, w->flag, w->pos, 0, (size_t)n, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(filt_scatter, dim3((n + 255) / 256), dim3(256), 0, st, w->pts[c], w->fac[c], w->src[c], n, w->flag, w->pos,
                     w->pts[o], w->fac[o], w->src[o], w->counts);
  FCHK(hipMemcpyAsync(w->host_pinned, w->counts, 4, hipMemcpyDeviceToHost, st));
  FCHK(hipStreamSynchronize(st));
  w->n = w->host_pinned[0];
  w->cur = o;
  return hipSuccess;
}

hipError_t filt_run_chain(FilterWorkspace* w, hipStream_t st, const smhip_filter_desc* chain, int nf, int* n_out) {
  if (!w || (nf > 0 && !chain)) return hipErrorInvalidValue;
  int k = 0;
  while (k < nf && w->n > 0) {
    const smhip_filter_desc& f = chain[k];
    const int n = w->n, gp = (n + 255) / 256;
    if (f.type == SMHIP_FILTER_RANGE || f.type == SMHIP_FILTER_AXIS_RANGE || f.type == SMHIP_FILTER_BOUNDING_BOX_REMOVAL) {
      PredGroup g{};
      while (k < nf && g.n < kMaxFused && (chain[k].type == SMHIP_FILTER_RANGE || chain[k].type == SMHIP_FILTER_AXIS_RANGE ||
                                           chain[k].type == SMHIP_FILTER_BOUNDING_BOX_REMOVAL)) {
        Pred& p = g.f[g.n++];
        p.type = chain[k].type; p.axis = chain[k].axis_index;
        for (int q = 0; q < 6; ++q) p.p[q] = chain[k].p[q];
        ++k;
      }
      hipLaunchKernelGGL(filt_flags, dim3(gp), dim3(256), 0, st, w->pts[w->cur], n, g, w->flag);
      FCHK(compact(w, st));
    } else if (f.type == SMHIP_FILTER_RANDOM_SAMPLER) {
      ++k;
      if ((double)f.p[0] > 0.999) continue;                                // filter_random_sample.cc:46-53: the float rate against the DOUBLE literal 0.999 (0.999f passes through)
      hipLaunchKernelGGL(filt_sample_flags, dim3(gp), dim3(256), 0, st, n, f.seed, f.p[0], w->flag);
      FCHK(compact(w, st));
    } else if (f.type == SMHIP_FILTER_VOXEL_GRID) {
      ++k;
      const int c = w->cur, o = 1 - c;
      FCHK(hipMemsetAsync(w->counts + 1, 0, 4, st));
      hipLaunchKernelGGL(filt_voxel_keys, dim3(gp), dim3(256), 0, st, w->pts[c], n, f.p[0], w->keys[0], w->idx[0], w->counts + 1);
      size_t bytes = w->tmp_bytes;
      FCHK(rocprim::radix_sort_pairs(w->tmp, bytes, w->keys[0], w->keys[1], w->idx[0], w->idx[1], (unsigned)n, 0, 64, st));
      hipLaunchKernelGGL(filt_voxel_heads, dim3(gp), dim3(256), 0, st, w->keys[1], n, w->flag);
      bytes = w->tmp_bytes;
      FCHK(rocprim::inclusive_scan(w->tmp, bytes, w->flag, w->pos, (size_t)n, rocprim::plus<int32_t>(), st));
      hipLaunchKernelGGL(filt_voxel_starts, dim3(gp), dim3(256), 0, st, w->flag, w->pos, n, w->start, w->counts);
      hipLaunchKernelGGL(filt_voxel_average, dim3(gp), dim3(256), 0, st, w->pts[c], w->idx[1], w->start, w->counts, n,
                         w->pts[o], w->fac[o], w->src[o]);
      FCHK(hipMemcpyAsync(w->host_pinned, w->counts, 8, hipMemcpyDeviceToHost, st));
      FCHK(hipStreamSynchronize(st));
      if (w->host_pinned[1] != 0) return hipErrorInvalidValue;             // a voxel index beyond +-2^20
      w->n = w->host_pinned[0];
      w->cur = o;
      w->has_index = false;
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (n_out) *n_out = w->n;
  return hipGetLastError();
}

const float4* filt_points(const FilterWorkspace* w) { return w->pts[w->cur]; }
const float* filt_factors(const FilterWorkspace* w) { return w->fac[w->cur]; }
const int32_t* filt_source_index(const FilterWorkspace* w) { return w->src[w->cur]; }
int filt_count(const FilterWorkspace* w) { return w->n; }
bool filt_has_index(const FilterWorkspace* w) { return w->has_index; }

}  // namespace smhip
