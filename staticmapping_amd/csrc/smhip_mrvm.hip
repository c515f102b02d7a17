// smhip_mrvm.hip -- static_map::MultiResolutionVoxelMap on the device: the probabilistic hit / miss voxel map with ray
// casting behind the reference's static-map output (/root/reference/builder/multi_resolution_voxel_map.{h,cc}, called per
// frame from builder/map_builder.cc:832-900; "mrvp using cuda or opencl" is on the reference's own to-do list, README.md:197).
//
// The reference's insert loop (multi_resolution_voxel_map.cc:76-126) is an OpenMP loop over the points whose probability
// updates race (:94 "not atomic"); what is reproduced here, exactly, is that loop executed in point order -- the result the
// reference gives when built without OpenMP.  In point order the state of a voxel after one cloud is a closed form:
//   * a voxel that EXISTED before the cloud and is hit first by point jmin takes one miss for every ray i < jmin that passes
//     through it (from jmin on its need_update flag is false, :88), then one hit for every point that ends in it;
//   * a voxel CREATED by the cloud is created by a hit with need_update false, so it takes its hits and no miss;
//   * a voxel the cloud does not hit takes one miss per ray through it.
// Misses all come before the hits and each is the same byte -> byte map (prob = uint8(clamp(odds^-1(odds(prob) + l)) * 256),
// :68-72), so the kernels only count: mrvm_hit (end voxels: insert-or-find in an open-addressing table, jmin, hit count,
// max intensity), mrvm_miss (one thread per ray walks VoxelCastingBresenham, common/math.cc:35-93, and counts a miss on
// every existing voxel the ray may still update), mrvm_apply (the two maps applied count times, from tables the host
// computes with the reference's own float / double expressions).  The first max_point_num_in_cell points of a voxel
// (:100-103, in point order) come from one radix sort of (voxel slot, point index).
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/smhip.h"

namespace {

constexpr int kTable = 256;                     // kTableSize, multi_resolution_voxel_map.h:47
constexpr uint8_t kUnknown = 128;               // :48
constexpr int kCoordBias = 1 << 20;             // voxel coordinates in (-2^20, 2^20): +-104 km at 0.1 m

struct MrvmDev {
  unsigned long long* keys;     // [T] 0 = empty, else 1 << 63 | (x + 2^20) << 42 | (y + 2^20) << 21 | (z + 2^20)
  uint8_t* prob;                // [T]
  int32_t* max_int;             // [T]
  int32_t* npts;                // [T]
  uint32_t* created;            // [T] insert number (epoch) that created the voxel
  float* pts;                   // [T][maxp][5]
  uint32_t* jmin;               // [T] smallest point index of the current cloud that ends in the voxel (0xffffffff: none)
  uint32_t* hits;               // [T]
  uint32_t* misses;             // [T]
  int32_t* touched;             // [T] voxels with a hit or a miss in the current cloud
  uint32_t* counters;           // 0 touched (this insert), 1 voxels in the table, 2 flags: bit 0 = a point of THIS insert lies beyond the
                                // coordinate range (reset per insert), bit 1 = table full (sticky: a voxel was lost), 3 output rows,
                                // 4 dump rows, 5 points of this insert skipped for their coordinates
  const uint8_t* tables;        // hit map [256], miss map [256]
  const float* cloud;           // [n][5] the current cloud (InnerPointType rows)
  int32_t* endslot;             // [n]
  unsigned long long* sort_keys[2];
  int32_t* run_start;           // [n]
  uint32_t tmask;
  int32_t maxp;
  uint32_t epoch;
  float res;
  float o[3];
};

__device__ __forceinline__ int voxel_of(float c, float step) {          // lround(floor(c / step)), common/math.cc:41-43
  return (int)lroundf(floorf(__fdiv_rn(c, step)));
}
__device__ __forceinline__ bool pack_key(int x, int y, int z, unsigned long long& key) {
  if (x <= -kCoordBias || x >= kCoordBias || y <= -kCoordBias || y >= kCoordBias || z <= -kCoordBias || z >= kCoordBias) return false;
  key = (1ull << 63) | ((unsigned long long)(x + kCoordBias) << 42) | ((unsigned long long)(y + kCoordBias) << 21) | (unsigned long long)(z + kCoordBias);
  return true;
}
__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32; k *= 0x94D049BB133111EBull; k ^= k >> 29;
  return (uint32_t)k;
}
__device__ __forceinline__ int find_slot(const MrvmDev& d, unsigned long long key) {
  uint32_t s = hash_key(key) & d.tmask;
  for (uint32_t probe = 0; probe <= d.tmask; ++probe) {
    const unsigned long long cur = __hip_atomic_load(&d.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int)s;
    if (cur == 0ull) return -1;
    s = (s + 1) & d.tmask;
  }
  return -1;
}

// Growth: every voxel of the old table into a larger one (between two inserts: the per-insert columns hits / misses / jmin are
// at rest).  A voxel's content does not depend on its slot, and every reader of the map orders by key, not by slot.
__global__ __launch_bounds__(256) void mrvm_rehash(MrvmDev o, MrvmDev nw) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > (size_t)o.tmask) return;
  const unsigned long long key = o.keys[s];
  if (key == 0ull) return;
  uint32_t t = hash_key(key) & nw.tmask;
  for (;;) {                                               // the new table holds at most half as many voxels as slots: ends
    if (atomicCAS(&nw.keys[t], 0ull, key) == 0ull) break;
    t = (t + 1) & nw.tmask;
  }
  nw.prob[t] = o.prob[s]; nw.max_int[t] = o.max_int[s]; nw.npts[t] = o.npts[s]; nw.created[t] = o.created[s];
  const int np = min(o.npts[s], o.maxp);
  const float* from = o.pts + (size_t)s * o.maxp * 5;
  float* to = nw.pts + (size_t)t * nw.maxp * 5;
  for (int k = 0; k < 5 * np; ++k) to[k] = from[k];
}

// per-insert reset: the touched list, the out-of-range flag and count of THIS cloud; "table full" stays (a lost voxel is lost)
__global__ void mrvm_begin(MrvmDev d) {
  d.counters[0] = 0;
  d.counters[2] &= 2u;
  d.counters[5] = 0;
}

// end voxels of the cloud: :86-104
__global__ __launch_bounds__(256) void mrvm_hit(MrvmDev d, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float* p = d.cloud + 5 * (size_t)j;
  const float x = p[0], y = p[1], z = p[2], inten = p[3];
  int slot = -1;
  unsigned long long key;
  if (isfinite(x) && isfinite(y) && isfinite(z)) {
    if (pack_key(voxel_of(x, d.res), voxel_of(y, d.res), voxel_of(z, d.res), key)) {
      uint32_t s = hash_key(key) & d.tmask;
      for (uint32_t probe = 0; probe <= d.tmask; ++probe) {
        const unsigned long long cur = __hip_atomic_load(&d.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) { slot = (int)s; break; }
        if (cur == 0ull) {
          const unsigned long long old = atomicCAS(&d.keys[s], 0ull, key);
          if (old == 0ull) { d.created[s] = d.epoch; atomicAdd(&d.counters[1], 1u); slot = (int)s; break; }
          if (old == key) { slot = (int)s; break; }
        }
        s = (s + 1) & d.tmask;
      }
      if (slot < 0) atomicOr(&d.counters[2], 2u);          // table full
    } else {
      atomicOr(&d.counters[2], 1u);                        // voxel coordinate out of range
      atomicAdd(&d.counters[5], 1u);
    }
  }
  d.endslot[j] = slot;
  if (slot < 0) return;
  atomicMin(&d.jmin[slot], (uint32_t)j);
  if (atomicAdd(&d.hits[slot], 1u) == 0u) d.touched[atomicAdd(&d.counters[0], 1u)] = slot;
  if (isfinite(inten)) atomicMax(&d.max_int[slot], (int)inten);                                  // :95-98
}

// the voxels on each ray, all but the last: :106-117
__global__ __launch_bounds__(256) void mrvm_miss(MrvmDev d, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || d.endslot[i] < 0) return;
  const float* p = d.cloud + 5 * (size_t)i;
  int x0 = voxel_of(d.o[0], d.res), y0 = voxel_of(d.o[1], d.res), z0 = voxel_of(d.o[2], d.res);
  const int xe = voxel_of(p[0], d.res), ye = voxel_of(p[1], d.res), ze = voxel_of(p[2], d.res);
  const int dx = abs(xe - x0), sx = x0 < xe ? 1 : -1;
  const int dy = abs(ye - y0), sy = y0 < ye ? 1 : -1;
  const int dz = abs(ze - z0), sz = z0 < ze ? 1 : -1;
  const int dm = max(dx, max(dy, dz));
  int ex = dm >> 1, ey = dm >> 1, ez = dm >> 1;
  for (int k = dm; k > 0; --k) {
    unsigned long long key;
    if (pack_key(x0, y0, z0, key)) {
      const int s = find_slot(d, key);
      // it exists (:109-110) and its need_update is still true (:111): it was there before this cloud and no earlier point of
      // the cloud ended in it
      if (s >= 0 && d.created[s] < d.epoch && d.jmin[s] > (uint32_t)i) {
        if (atomicAdd(&d.misses[s], 1u) == 0u && d.hits[s] == 0u) d.touched[atomicAdd(&d.counters[0], 1u)] = s;
      }
    }
    ex -= dx; if (ex < 0) { ex += dm; x0 += sx; }
    ey -= dy; if (ey < 0) { ey += dm; y0 += sy; }
    ez -= dz; if (ez < 0) { ez += dm; z0 += sz; }
  }
}

__global__ __launch_bounds__(256) void mrvm_apply(MrvmDev d) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.counters[0]) return;
  const int s = d.touched[t];
  uint8_t p = d.prob[s];
  uint32_t nm = d.misses[s], nh = d.hits[s];
  const uint8_t* hit = d.tables;
  const uint8_t* miss = d.tables + kTable;
  for (; nm > 0 && miss[p] != p; --nm) p = miss[p];         // a fixed point of the map stays one
  for (; nh > 0 && hit[p] != p; --nh) p = hit[p];
  d.prob[s] = p;
  d.misses[s] = 0; d.hits[s] = 0; d.jmin[s] = 0xffffffffu;
}

// (slot, point index) of every point that ended in a voxel, for the radix sort
__global__ __launch_bounds__(256) void mrvm_point_keys(MrvmDev d, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = d.endslot[j];
  d.sort_keys[0][j] = s >= 0 ? ((unsigned long long)(uint32_t)s << 32) | (uint32_t)j : ~0ull;
}
__global__ __launch_bounds__(256) void mrvm_run_heads(MrvmDev d, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long k = d.sort_keys[1][r];
  d.run_start[r] = (r == 0 || (d.sort_keys[1][r - 1] >> 32) != (k >> 32)) ? r : 0;
}
// the first (max_point_num_in_cell - size()) points of every voxel's run, in point order: :100-103
__global__ __launch_bounds__(256) void mrvm_store_points(MrvmDev d, int n, const int32_t* start) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long k = d.sort_keys[1][r];
  if (k == ~0ull) return;
  const int s = (int)(k >> 32), j = (int)(k & 0xffffffffu);
  const int at = d.npts[s] + (r - start[r]);
  if (at >= d.maxp) return;
  const float* p = d.cloud + 5 * (size_t)j;
  float* out = d.pts + ((size_t)s * d.maxp + at) * 5;
  for (int c = 0; c < 5; ++c) out[c] = p[c];
}
__global__ __launch_bounds__(256) void mrvm_update_counts(MrvmDev d, int n, const int32_t* start) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long k = d.sort_keys[1][r];
  if (k == ~0ull) return;
  const bool last = r == n - 1 || (d.sort_keys[1][r + 1] >> 32) != (k >> 32);
  if (!last) return;
  const int s = (int)(k >> 32);
  d.npts[s] = min(d.maxp, d.npts[s] + (r - start[r] + 1));
}

// OutputToPointCloud, :125-216.  flags bit 0 = settings_.output_average (one row per voxel: the float sums of the stored points in
// their order, divided by float(size)), bit 1 = the PointXYZRGB overload (4th column = the bits of (255 << 24 | g << 16 | g << 8 | g) with
// g = min(255, uint32(max_intensity * 1.4)), :181-186), else PointXYZI (4th column = the voxel's max intensity when
// use_max_intensity, else the point's own -- 0 for an averaged point, whose intensity is never assigned, :148-151)
__global__ __launch_bounds__(256) void mrvm_output(MrvmDev d, uint8_t thr, int use_max, int flags, float* xyzi, int capacity) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > d.tmask || d.keys[s] == 0ull || d.prob[s] < thr) return;
  const int c = d.npts[s];
  if (c <= 0) return;
  const bool average = flags & 1, rgb = flags & 2;
  float grey = 0.f;
  if (rgb) {
    uint32_t g = (uint32_t)d.max_int[s];
    g = (uint32_t)((double)g * 1.4);                                          // intensity *= 1.4 on a uint32_t
    if (g > 255u) g = 255u;
    grey = __uint_as_float(0xff000000u | (g << 16) | (g << 8) | g);            // a = 255: what pcl::PointXYZRGB's constructor (PCL >= 1.8) leaves in the byte the reference never assigns
  }
  const uint32_t base = atomicAdd(&d.counters[3], average ? 1u : (uint32_t)c);
  if (average) {
    if ((long long)base >= capacity) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int k = 0; k < c; ++k) {
      const float* p = d.pts + ((size_t)s * d.maxp + k) * 5;
      ax = __fadd_rn(ax, p[0]); ay = __fadd_rn(ay, p[1]); az = __fadd_rn(az, p[2]);
    }
    const float size = (float)c;
    float* o = xyzi + 4 * (size_t)base;
    o[0] = __fdiv_rn(ax, size); o[1] = __fdiv_rn(ay, size); o[2] = __fdiv_rn(az, size);
    o[3] = rgb ? grey : (use_max ? (float)d.max_int[s] : 0.f);
    return;
  }
  for (int k = 0; k < c; ++k) {
    if ((long long)base + k >= capacity) return;
    const float* p = d.pts + ((size_t)s * d.maxp + k) * 5;
    float* o = xyzi + 4 * ((size_t)base + k);
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = rgb ? grey : (use_max ? (float)d.max_int[s] : p[3]);
  }
}
// every voxel, for the parity tests
__global__ __launch_bounds__(256) void mrvm_dump(MrvmDev d, int32_t* keys3, uint8_t* prob, int32_t* max_int, int32_t* npts, float* pts, int capacity) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > d.tmask || d.keys[s] == 0ull) return;
  const uint32_t at = atomicAdd(&d.counters[4], 1u);
  if ((int)at >= capacity) return;
  const unsigned long long k = d.keys[s];
  keys3[3 * at] = (int)((k >> 42) & 0x1fffffull) - kCoordBias;
  keys3[3 * at + 1] = (int)((k >> 21) & 0x1fffffull) - kCoordBias;
  keys3[3 * at + 2] = (int)(k & 0x1fffffull) - kCoordBias;
  prob[at] = d.prob[s]; max_int[at] = d.max_int[s]; npts[at] = d.npts[s];
  for (int c = 0; c < d.npts[s] * 5; ++c) pts[(size_t)at * d.maxp * 5 + c] = d.pts[(size_t)s * d.maxp * 5 + c];
}

float clampf(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }             // common::Clamp, math.h:66-75
// ProbabilityToOdd / OddToProbability with the reference's float / double mix (header :133-139)
float prob_to_odd(float p) { return static_cast<float>(std::log(p / (1. - p))); }
float odd_to_prob(float odd) { return static_cast<float>(1. - 1. / (1. + std::exp(odd))); }

}  // namespace

struct smhip_mrvm_context {
  int device = 0;
  hipStream_t stream = nullptr;
  MrvmDev d{};
  smhip_mrvm_settings set{};
  size_t T = 0;
  int max_cloud = 0;
  float* cloud_dev = nullptr;
  float* stage = nullptr;               // pinned
  uint32_t* counters_host = nullptr;    // pinned
  uint8_t* tables_dev = nullptr;
  int32_t* scan_out = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_bytes = 0;
  std::vector<void*> allocs;
  std::string err;
  int last_skipped = 0;                 // points of the last insert skipped for their coordinates
  size_t voxels = 0;                    // voxels in the table after the last insert
  bool voxels_stale = false;            // an insert was started and did not reach the point where `voxels` is refreshed
  int max_table_log2 = 28;              // growth stops here (smhip_mrvm_set_max_table_log2)
  int growths = 0;
};

// the table's columns for T slots, initialised empty; false (nothing kept) when the device refuses the memory
static bool mrvm_alloc_table(smhip_mrvm_context* h, size_t T, MrvmDev* d, std::vector<void*>* got) {
  const size_t P = (size_t)h->set.max_point_num_in_cell;
  bool ok = true;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) == hipSuccess) got->push_back(*p); else ok = false; };
  A((void**)&d->keys, T * 8); A((void**)&d->prob, T); A((void**)&d->max_int, T * 4); A((void**)&d->npts, T * 4); A((void**)&d->created, T * 4);
  A((void**)&d->pts, T * P * 5 * 4); A((void**)&d->jmin, T * 4); A((void**)&d->hits, T * 4); A((void**)&d->misses, T * 4); A((void**)&d->touched, T * 4);
  ok = ok && hipMemsetAsync(d->keys, 0, T * 8, h->stream) == hipSuccess && hipMemsetAsync(d->prob, kUnknown, T, h->stream) == hipSuccess &&
       hipMemsetAsync(d->max_int, 0, T * 4, h->stream) == hipSuccess && hipMemsetAsync(d->npts, 0, T * 4, h->stream) == hipSuccess &&
       hipMemsetAsync(d->created, 0, T * 4, h->stream) == hipSuccess && hipMemsetAsync(d->jmin, 0xff, T * 4, h->stream) == hipSuccess &&
       hipMemsetAsync(d->hits, 0, T * 4, h->stream) == hipSuccess && hipMemsetAsync(d->misses, 0, T * 4, h->stream) == hipSuccess;
  if (!ok) { for (void* p : *got) (void)hipFree(p); got->clear(); (void)hipGetLastError(); }
  d->tmask = (uint32_t)(T - 1);
  return ok;
}

// The reference's map grows without bound (std::map of voxels); this table doubles until the voxels it holds plus the points of
// the coming cloud (an insert creates at most one voxel per point) fill at most half of it, up to 2^max_table_log2 slots.
// Called between inserts.  A refused allocation leaves the old table in place (the insert then reports what it always did).
static void mrvm_grow_for(smhip_mrvm_context* h, int n) {
  const size_t need = h->voxels + (size_t)n;
  if (need * 10 <= h->T * 6) return;
  size_t T = h->T;
  const size_t cap = (size_t)1 << h->max_table_log2;
  while (T < cap && need * 2 > T) T <<= 1;
  if (T <= h->T) return;
  MrvmDev nw = h->d;
  std::vector<void*> got;
  if (!mrvm_alloc_table(h, T, &nw, &got)) return;
  hipLaunchKernelGGL(mrvm_rehash, dim3((unsigned)((h->T + 255) / 256)), dim3(256), 0, h->stream, h->d, nw);
  if (hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) { for (void* p : got) (void)hipFree(p); return; }
  void* old[10] = {h->d.keys, h->d.prob, h->d.max_int, h->d.npts, h->d.created, h->d.pts, h->d.jmin, h->d.hits, h->d.misses, h->d.touched};
  for (void* p : old) {
    (void)hipFree(p);
    h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), p), h->allocs.end());
  }
  for (void* p : got) h->allocs.push_back(p);
  h->d = nw;
  h->T = T;
  h->growths++;
}

#define MCHK(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return SMHIP_ERR_HIP; } } while (0)

extern "C" {

void smhip_mrvm_default_settings(smhip_mrvm_settings* s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->high_resolution = 0.1f; s->hit_prob = 0.55f; s->miss_prob = 0.48f; s->z_offset = 0.f;      // MrvmSettings, header :54-65
  s->prob_threshold = 0.6f; s->max_point_num_in_cell = 10; s->use_max_intensity = 1;
}

smhip_status smhip_mrvm_create(int device, int table_log2, int max_cloud_points, const smhip_mrvm_settings* settings, smhip_mrvm_handle* out) {
  if (!out || !settings || table_log2 < 10 || table_log2 > 28 || max_cloud_points < 1) return SMHIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (settings->max_point_num_in_cell <= 0 || !(settings->high_resolution > 0.f)) return SMHIP_ERR_INVALID_ARGUMENT;   // CHECK_GT(max_point_num_in_cell, 0), .cc:48
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SMHIP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SMHIP_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return SMHIP_ERR_NO_DEVICE;
  smhip_mrvm_context* h = new smhip_mrvm_context();
  h->device = device; h->set = *settings; h->T = (size_t)1 << table_log2; h->max_cloud = max_cloud_points;
  h->set.hit_prob = clampf(settings->hit_prob, 0.501f, 0.9f);                                    // Initialise, .cc:50-52
  h->set.miss_prob = clampf(settings->miss_prob, 0.1f, 0.499f);
  bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) == hipSuccess) h->allocs.push_back(*p); else ok = false; };
  MrvmDev& d = h->d;
  const size_t T = h->T, N = (size_t)max_cloud_points;
  { std::vector<void*> got; ok = ok && mrvm_alloc_table(h, T, &d, &got); for (void* p : got) h->allocs.push_back(p); }
  A((void**)&d.counters, 64); A((void**)&h->tables_dev, 2 * kTable); A((void**)&h->cloud_dev, N * 5 * 4); A((void**)&d.endslot, N * 4);
  A((void**)&d.sort_keys[0], N * 8); A((void**)&d.sort_keys[1], N * 8); A((void**)&d.run_start, N * 4); A((void**)&h->scan_out, N * 4);
  if (ok) {
    size_t b1 = 0, b2 = 0;
    (void)rocprim::radix_sort_keys(nullptr, b1, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned)N, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b2, (const int32_t*)nullptr, (int32_t*)nullptr, N, rocprim::maximum<int32_t>(), (hipStream_t)0);
    h->sort_bytes = std::max(b1, b2) + 256;
    A(&h->sort_tmp, h->sort_bytes);
  }
  ok = ok && hipHostMalloc((void**)&h->stage, N * 5 * 4) == hipSuccess && hipHostMalloc((void**)&h->counters_host, 64) == hipSuccess;
  if (ok) {
    ok = hipMemsetAsync(d.counters, 0, 64, h->stream) == hipSuccess;
    // the byte -> byte maps of one hit / one miss, with the reference's expressions: odds_table_ (.cc:41-43), update_prob (:68-72)
    uint8_t tab[2 * kTable];
    const float hit_log_odd = prob_to_odd(h->set.hit_prob), miss_log_odd = prob_to_odd(h->set.miss_prob);
    for (int p = 0; p < kTable; ++p) {
      const float base = prob_to_odd(static_cast<float>(p) / kTable);
      float odd = base; odd += hit_log_odd;
      tab[p] = static_cast<uint8_t>(clampf(odd_to_prob(odd), 0.1f, 0.9f) * kTable);
      odd = base; odd += miss_log_odd;
      tab[kTable + p] = static_cast<uint8_t>(clampf(odd_to_prob(odd), 0.1f, 0.9f) * kTable);
    }
    ok = ok && hipMemcpyAsync(h->tables_dev, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
  }
  if (!ok) { smhip_mrvm_destroy(h); return SMHIP_ERR_HIP; }
  d.tables = h->tables_dev; d.cloud = h->cloud_dev;
  d.tmask = (uint32_t)(T - 1); d.maxp = settings->max_point_num_in_cell; d.epoch = 0; d.res = settings->high_resolution;
  *out = h;
  return SMHIP_OK;
}

smhip_status smhip_mrvm_destroy(smhip_mrvm_handle h) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->stage) (void)hipHostFree(h->stage);
  if (h->counters_host) (void)hipHostFree(h->counters_host);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return SMHIP_OK;
}

const char* smhip_mrvm_last_error(smhip_mrvm_handle h) { return h ? h->err.c_str() : "null handle"; }

void smhip_mrvm_set_offset_z(smhip_mrvm_handle h, float offset) { if (h) h->set.z_offset = offset; }   // SetOffsetZ, .cc:55-57

smhip_status smhip_mrvm_insert_f32(smhip_mrvm_handle h, const float* points, int stride_floats, int n, const float origin[3]) {
  if (!h || !origin) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!points || n <= 0) { h->err = "cloud is empty."; return SMHIP_ERR_INVALID_ARGUMENT; }             // PRINT_ERROR + return, .cc:61-64
  if (stride_floats < 4) { h->err = "rows need x y z intensity"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->max_cloud) { h->err = "cloud larger than max_cloud_points"; return SMHIP_ERR_CAPACITY; }
  {   // every ray starts at the origin: a non-finite or far-away one would walk ~2^21 voxels per ray for nothing.  Checked before
      // anything is touched, so a refused cloud leaves the map as it was.
    const float oz = origin[2] + h->set.z_offset;
    const float lim = (float)(kCoordBias - 2) * h->set.high_resolution;
    if (!(std::isfinite(origin[0]) && std::isfinite(origin[1]) && std::isfinite(oz)) ||
        std::fabs(origin[0]) >= lim || std::fabs(origin[1]) >= lim || std::fabs(oz) >= lim) {
      h->err = "origin is not finite or lies beyond +-2^20 voxels: cloud refused";
      return SMHIP_ERR_INVALID_ARGUMENT;
    }
  }
  h->err.clear();
  h->last_skipped = 0;
  MCHK(h, hipSetDevice(h->device));
  MCHK(h, hipStreamSynchronize(h->stream));
  if (h->voxels_stale) {                                    // the previous insert ended early: its voxels are in the table, not in `voxels`
    MCHK(h, hipMemcpy(h->counters_host, h->d.counters, 16, hipMemcpyDeviceToHost));
    h->voxels = h->counters_host[1];
  }
  h->voxels_stale = true;
  mrvm_grow_for(h, n);                                      // room for the voxels this cloud can add, like the reference's std::map
  for (int i = 0; i < n; ++i) {
    const float* r = points + (size_t)stride_floats * i;
    float* o = h->stage + 5 * (size_t)i;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = stride_floats > 4 ? r[4] : 0.f;
  }
  MrvmDev& d = h->d;
  ++d.epoch;
  d.o[0] = origin[0]; d.o[1] = origin[1]; d.o[2] = origin[2] + h->set.z_offset;                         // .cc:66-67
  MCHK(h, hipMemcpyAsync(h->cloud_dev, h->stage, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(mrvm_begin, dim3(1), dim3(1), 0, h->stream, d);                                    // touched count, this cloud's flags
  const dim3 g((n + 255) / 256), b(256);
  hipLaunchKernelGGL(mrvm_hit, g, b, 0, h->stream, d, n);
  hipLaunchKernelGGL(mrvm_miss, g, b, 0, h->stream, d, n);
  // the points kept per voxel: before the hit counts are consumed? they are independent of them -- sort (slot, index) pairs
  hipLaunchKernelGGL(mrvm_point_keys, g, b, 0, h->stream, d, n);
  size_t bytes = h->sort_bytes;
  MCHK(h, rocprim::radix_sort_keys(h->sort_tmp, bytes, d.sort_keys[0], d.sort_keys[1], (unsigned)n, 0, 64, h->stream));
  hipLaunchKernelGGL(mrvm_run_heads, g, b, 0, h->stream, d, n);
  bytes = h->sort_bytes;
  MCHK(h, rocprim::inclusive_scan(h->sort_tmp, bytes, d.run_start, h->scan_out, (size_t)n, rocprim::maximum<int32_t>(), h->stream));
  hipLaunchKernelGGL(mrvm_store_points, g, b, 0, h->stream, d, n, h->scan_out);
  hipLaunchKernelGGL(mrvm_update_counts, g, b, 0, h->stream, d, n, h->scan_out);
  MCHK(h, hipMemcpyAsync(h->counters_host, d.counters, 32, hipMemcpyDeviceToHost, h->stream));
  MCHK(h, hipStreamSynchronize(h->stream));
  const uint32_t touched = h->counters_host[0];
  if (touched > 0) hipLaunchKernelGGL(mrvm_apply, dim3((touched + 255) / 256), b, 0, h->stream, d);
  MCHK(h, hipGetLastError());
  // What is reported after the cloud has been applied.  Only a LOST voxel is an error (sticky: the map is incomplete from then
  // on -- the table grows between inserts like the reference's map, until max_table_log2 or the device's memory ends it).  Points beyond the coordinate
  // range were skipped, the rest of the cloud is in the map: status OK, the count in smhip_mrvm_last_skipped, the text in
  // smhip_mrvm_last_error.  The same for a table that is getting full.
  h->last_skipped = (int)h->counters_host[5];
  h->voxels = h->counters_host[1];
  h->voxels_stale = false;
  if (h->counters_host[2] & 2u) {
    h->err = "voxel table full: at least one voxel of this or an earlier cloud was dropped (the rest was applied); the table could not grow "
             "(smhip_mrvm_set_max_table_log2, or device memory)";
    return SMHIP_ERR_CAPACITY;
  }
  if (h->counters_host[2] & 1u) h->err = "warning: " + std::to_string(h->last_skipped) + " point(s) beyond +-2^20 voxels skipped, the rest of the cloud applied";
  else if ((size_t)h->counters_host[1] * 10 > h->T * 7) h->err = "warning: voxel table more than 70 % full and at its largest size (smhip_mrvm_set_max_table_log2)";
  return SMHIP_OK;
}

smhip_status smhip_mrvm_set_max_table_log2(smhip_mrvm_handle h, int max_table_log2) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (max_table_log2 < 10 || max_table_log2 > 28) { h->err = "max_table_log2 must be in 10..28"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->err.clear();
  h->max_table_log2 = max_table_log2;
  return SMHIP_OK;
}

int smhip_mrvm_table_log2(smhip_mrvm_handle h) {
  if (!h) return 0;
  int l = 0;
  while (((size_t)1 << l) < h->T) ++l;
  return l;
}

smhip_status smhip_mrvm_voxel_count(smhip_mrvm_handle h, int* n) {
  if (!h || !n) { if (h) h->err = "null output pointer"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->err.clear();
  MCHK(h, hipSetDevice(h->device));
  MCHK(h, hipMemcpyAsync(h->counters_host, h->d.counters, 16, hipMemcpyDeviceToHost, h->stream));
  MCHK(h, hipStreamSynchronize(h->stream));
  *n = (int)h->counters_host[1];
  return SMHIP_OK;
}

smhip_status smhip_mrvm_output_ex(smhip_mrvm_handle h, float threshold, int flags, float* rows, int capacity, int* n_out) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!n_out || (capacity > 0 && !rows) || (flags & ~3)) { h->err = "output: null pointer or unknown flag"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->err.clear();                                           // (an insert's warning text does not outlive the next call)
  MCHK(h, hipSetDevice(h->device));
  float* dev = nullptr;
  if (capacity > 0) MCHK(h, hipMalloc((void**)&dev, sizeof(float) * 4 * (size_t)capacity));
  MCHK(h, hipMemsetAsync(h->d.counters + 3, 0, 4, h->stream));
  const uint8_t thr = static_cast<uint8_t>(threshold * kTable);                                         // .cc:132
  hipLaunchKernelGGL(mrvm_output, dim3((unsigned)((h->T + 255) / 256)), dim3(256), 0, h->stream, h->d, thr, h->set.use_max_intensity, flags, dev, capacity);
  MCHK(h, hipMemcpyAsync(h->counters_host, h->d.counters, 32, hipMemcpyDeviceToHost, h->stream));
  MCHK(h, hipStreamSynchronize(h->stream));
  *n_out = (int)h->counters_host[3];
  if (capacity > 0) {
    const hipError_t e = hipMemcpy(rows, dev, sizeof(float) * 4 * (size_t)std::min(capacity, *n_out), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return SMHIP_ERR_HIP; }
  }
  return SMHIP_OK;
}

smhip_status smhip_mrvm_output(smhip_mrvm_handle h, float threshold, float* xyzi, int capacity, int* n_out) {
  return smhip_mrvm_output_ex(h, threshold, 0, xyzi, capacity, n_out);
}

smhip_status smhip_mrvm_last_skipped(smhip_mrvm_handle h, int* n) {
  if (!h || !n) return SMHIP_ERR_INVALID_ARGUMENT;
  *n = h->last_skipped;
  return SMHIP_OK;
}

smhip_status smhip_mrvm_dump(smhip_mrvm_handle h, int32_t* keys3, uint8_t* prob, int32_t* max_intensity, int32_t* npoints, float* points5, int capacity, int* n_out) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!n_out || capacity < 1 || !keys3 || !prob || !max_intensity || !npoints || !points5) { h->err = "dump: null pointer or no capacity"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->err.clear();
  MCHK(h, hipSetDevice(h->device));
  const size_t C = (size_t)capacity, P = (size_t)h->d.maxp;
  int32_t *dk = nullptr, *dm = nullptr, *dn = nullptr; uint8_t* dp = nullptr; float* dq = nullptr;
  bool ok = hipMalloc((void**)&dk, C * 12) == hipSuccess && hipMalloc((void**)&dp, C) == hipSuccess && hipMalloc((void**)&dm, C * 4) == hipSuccess &&
            hipMalloc((void**)&dn, C * 4) == hipSuccess && hipMalloc((void**)&dq, C * P * 20) == hipSuccess;
  smhip_status st = SMHIP_OK;
  if (ok) {
    ok = hipMemsetAsync(h->d.counters + 4, 0, 4, h->stream) == hipSuccess && hipMemsetAsync(dq, 0, C * P * 20, h->stream) == hipSuccess;
    hipLaunchKernelGGL(mrvm_dump, dim3((unsigned)((h->T + 255) / 256)), dim3(256), 0, h->stream, h->d, dk, dp, dm, dn, dq, capacity);
    ok = ok && hipMemcpyAsync(h->counters_host, h->d.counters, 32, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
    if (ok) {
      *n_out = (int)h->counters_host[4];
      const size_t m = (size_t)std::min(capacity, *n_out);
      ok = hipMemcpy(keys3, dk, m * 12, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(prob, dp, m, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(max_intensity, dm, m * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(npoints, dn, m * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(points5, dq, m * P * 20, hipMemcpyDeviceToHost) == hipSuccess;
    }
  }
  if (!ok) { h->err = "dump failed (allocation or copy)"; st = SMHIP_ERR_HIP; }
  (void)hipFree(dk); (void)hipFree(dp); (void)hipFree(dm); (void)hipFree(dn); (void)hipFree(dq);
  return st;
}

}  // extern "C"
