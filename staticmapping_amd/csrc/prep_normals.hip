// prep_normals.hip -- EigenPointCloud::CalculateNormals on the device.
//
// Reference: /root/reference/builder/data/cloud_types.cc:347-368 (driver), :105-144 (BuildNormals: kd-box
// split on the widest tracked-bbox axis by std::nth_element until <= 7 points), :73-103 (leaf: mean point,
// unconstrained least-squares normal, rank test).  The caller of the registrator runs it on every new
// key-frame right before SetInputTarget (builder/map_builder.cc:286,389), so it is the step immediately
// before the hot path; on the host it costs ~30 ms per 120 k-point scan, 200x the GPU alignment itself.
//
// Level-synchronous formulation: all nodes of one tree level are split together.  Points of a node occupy a
// contiguous segment of `order`; one stable radix sort (rocPRIM building block) of the composite key
// (segment start << 32 | order-preserving float bits of the node's cut coordinate) sorts every active
// segment along its own axis; the split index count - count / 2 and the cut value are then read off.
// Exact medians, so the partition equals nth_element's up to ties in the cut coordinate (which
// std::nth_element leaves unspecified as well).
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>

#include "prep_normals.h"
#include "kd_median_tree.h"

namespace smhip {

namespace {

constexpr int kLeafMax = 7;              // kNormalEstimationKnn, cloud_types.cc:38
constexpr int kMaxScans = 512;           // scans one batched CalculateNormals call may hold

struct KdNode {
  int32_t start, count;
  float lo[3], hi[3];                    // tracked bounding box (exact: every value is a point coordinate)
  int32_t dim;
  int32_t left;                          // points going to the left child
};

__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct KdLeaf { int32_t start, count; };   // a node that will not be split further

// S scans are processed as ONE forest: scan s owns positions [prefix[s], prefix[s + 1]) and is one root.
struct ScanSet {
  int32_t S;
  const int32_t* offset;     // [S] first element of scan s in `raw`
  const int32_t* prefix;     // [S + 1] positions
};

__device__ __forceinline__ int scan_of_pos(const ScanSet& ss, int pos) {
  int lo = 0, hi = ss.S - 1;                       // last s with prefix[s] <= pos
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ss.prefix[mid] <= pos) lo = mid; else hi = mid - 1; }
  return lo;
}

__global__ __launch_bounds__(1024) void kd_bbox(const float4* raw, ScanSet ss, float* out) {   // one block per scan
  const int sc = blockIdx.x;
  const float4* p0 = raw + ss.offset[sc];
  const int n = ss.prefix[sc + 1] - ss.prefix[sc];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float4 p = p0[i];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s[16][6];
  for (int c = 0; c < 3; ++c)
    for (int off = 32; off > 0; off >>= 1) { mn[c] = fminf(mn[c], __shfl_down(mn[c], off, 64)); mx[c] = fmaxf(mx[c], __shfl_down(mx[c], off, 64)); }
  if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = mn[c]; s[threadIdx.x >> 6][3 + c] = mx[c]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    out[8 * sc + threadIdx.x] = v;
  }
}

__global__ void kd_init(ScanSet ss, int total, int32_t* order, int32_t* seg, int32_t* node_at, KdNode* nodes, int32_t* counts,
                        const float* bbox, KdLeaf* leaves) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos < total) {
    const int sc = scan_of_pos(ss, pos);
    order[pos] = ss.offset[sc] + (pos - ss.prefix[sc]);
    seg[pos] = ss.prefix[sc];
    node_at[pos] = -1;
  }
  if (pos == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 0; }
}

__global__ void kd_roots(ScanSet ss, int32_t* node_at, KdNode* nodes, int32_t* counts, const float* bbox, KdLeaf* leaves) {
  const int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc >= ss.S) return;
  const int n = ss.prefix[sc + 1] - ss.prefix[sc];
  if (n <= 0) return;
  if (n > kLeafMax) {
    KdNode r;
    r.start = ss.prefix[sc]; r.count = n; r.dim = 0; r.left = 0;
    for (int c = 0; c < 3; ++c) { r.lo[c] = bbox[8 * sc + c]; r.hi[c] = bbox[8 * sc + 3 + c]; }
    const int slot = atomicAdd(&counts[0], 1);
    nodes[slot] = r;
    node_at[r.start] = slot;
  } else {
    const int slot = atomicAdd(&counts[2], 1);
    leaves[slot].start = ss.prefix[sc]; leaves[slot].count = n;
  }
}

__global__ void kd_choose_dim(KdNode* nodes, const int32_t* counts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= counts[0]) return;
  KdNode& nd = nodes[k];
  // ArgMax of cloud_types.cc:41-56: starts from (index 0, value 0.0), strict ">"
  double best = 0.0; int dim = 0;
  for (int c = 0; c < 3; ++c) { const double e = (double)nd.hi[c] - (double)nd.lo[c]; if (e > best) { best = e; dim = c; } }
  nd.dim = dim;
  const int right = nd.count / 2;                        // :118
  nd.left = nd.count - right;
}

__global__ void kd_keys(const float4* raw, int n, const int32_t* order, const int32_t* seg, const int32_t* node_at,
                        const KdNode* nodes, unsigned long long* keys) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const int st = seg[pos];
  const int k = node_at[st];
  uint32_t lowbits;
  if (k >= 0) {
    const float4 p = raw[order[pos]];
    const int dim = nodes[k].dim;
    lowbits = ordered_bits(dim == 0 ? p.x : (dim == 1 ? p.y : p.z));
  } else {
    lowbits = (uint32_t)(pos - st);                       // finished segment: keep its order
  }
  keys[pos] = ((unsigned long long)(uint32_t)st << 32) | lowbits;
}

// ascending sort of a few point indices in place (insertion sort; heap sort for a long run of equal coordinates)
__device__ void sort_indices(int32_t* a, int n) {
  if (n <= 24) {
    for (int i = 1; i < n; ++i) {
      const int32_t v = a[i];
      int j = i;
      while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; --j; }
      a[j] = v;
    }
    return;
  }
  auto sift = [&](int root, int end) {
    for (;;) {
      int c = 2 * root + 1;
      if (c >= end) return;
      if (c + 1 < end && a[c + 1] > a[c]) ++c;
      if (a[root] >= a[c]) return;
      const int32_t t = a[root]; a[root] = a[c]; a[c] = t;
      root = c;
    }
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
  for (int e = n - 1; e > 0; --e) { const int32_t t = a[0]; a[0] = a[e]; a[e] = t; sift(0, e); }
}

// children of every active node; leaves are appended to the leaf list
__global__ void kd_split(const float4* raw, int32_t* order, const KdNode* nodes, KdNode* next, int32_t* counts,
                         int32_t* node_at_next, KdLeaf* leaves) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= counts[0]) return;
  const KdNode nd = nodes[k];
  // Points ON the median value: std::nth_element leaves open which of them go left; the rule of this code base (the forest of
  // kd_median_tree.h has it too, so a scan's leaves do not depend on the size of the batch it is prepared in) is "the smallest
  // indices".  The level's stable sort left equal coordinates in the previous level's order: when a run of them straddles the
  // split position, that run is put in index order.  (Tie-free data never enters the branch.)
  auto key_at = [&](int pos) {
    const float4 p = raw[order[pos]];
    return ordered_bits(nd.dim == 0 ? p.x : (nd.dim == 1 ? p.y : p.z));
  };
  const int m = nd.start + nd.left;
  if (nd.left > 0 && nd.left < nd.count) {
    const uint32_t km = key_at(m);
    if (key_at(m - 1) == km) {
      int a = m - 1, b = m + 1;
      while (a > nd.start && key_at(a - 1) == km) --a;
      while (b < nd.start + nd.count && key_at(b) == km) ++b;
      sort_indices(order + a, b - a);
    }
  }
  const float4 pc = raw[order[m]];                         // the nth element, :128
  const float cut = nd.dim == 0 ? pc.x : (nd.dim == 1 ? pc.y : pc.z);
  KdNode ch[2];
  ch[0].start = nd.start; ch[0].count = nd.left;
  ch[1].start = nd.start + nd.left; ch[1].count = nd.count - nd.left;
  for (int c = 0; c < 3; ++c) { ch[0].lo[c] = nd.lo[c]; ch[0].hi[c] = nd.hi[c]; ch[1].lo[c] = nd.lo[c]; ch[1].hi[c] = nd.hi[c]; }
  ch[0].hi[nd.dim] = cut;                                  // :132-133
  ch[1].lo[nd.dim] = cut;                                  // :135-136
  for (int s = 0; s < 2; ++s) {
    ch[s].dim = 0; ch[s].left = 0;
    if (ch[s].count > kLeafMax) {
      const int slot = atomicAdd(&counts[1], 1);
      next[slot] = ch[s];
      node_at_next[ch[s].start] = slot;
    } else {
      const int slot = atomicAdd(&counts[2], 1);
      leaves[slot].start = ch[s].start; leaves[slot].count = ch[s].count;
      node_at_next[ch[s].start] = -1;
    }
  }
}

__global__ void kd_update_seg(int n, const int32_t* seg, const int32_t* node_at, const KdNode* nodes, int32_t* seg_next) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const int st = seg[pos];
  const int k = node_at[st];
  int out = st;
  if (k >= 0 && pos - st >= nodes[k].left) out = st + nodes[k].left;
  seg_next[pos] = out;
}

__global__ void kd_advance(int32_t* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { counts[0] = counts[1]; counts[1] = 0; }
}

__device__ int rank3_sym(const double* C) {
  double A[9];
  for (int i = 0; i < 9; ++i) A[i] = C[i];
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = A[3 * k + p], akq = A[3 * k + q]; A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = A[3 * p + k], aqk = A[3 * q + k]; A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk; }
      }
  }
  const double w[3] = {fabs(A[0]), fabs(A[4]), fabs(A[8])};
  const double wmax = fmax(w[0], fmax(w[1], w[2]));
  int r = 0;
  for (int i = 0; i < 3; ++i) if (w[i] > 2.220446049250313e-16 * 3 * wmax) ++r;
  return r;
}

// one thread per leaf: cloud_types.cc:73-103
__global__ void kd_leaf_normals(const float4* raw, ScanSet ss, const int32_t* order, const KdLeaf* leaves, const int32_t* counts,
                                float4* leaf_p, float4* leaf_n, unsigned long long* leaf_key, int32_t* leaf_id) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= counts[2]) return;
  const KdLeaf lf = leaves[l];
  double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  int kmin = 0x7fffffff;
  // the leaf's points in index order, whatever order the tree builder left them in (the one-workgroup forest places them by
  // atomic counters): the sums below are then the same bits in every run
  int ids[kLeafMax];
#pragma unroll
  for (int i = 0; i < kLeafMax; ++i) ids[i] = i < lf.count ? order[lf.start + i] : 0x7fffffff;
#pragma unroll
  for (int a = 1; a < kLeafMax; ++a)
#pragma unroll
    for (int c = kLeafMax - 1; c >= a; --c) { const int lo = min(ids[c - 1], ids[c]), hi = max(ids[c - 1], ids[c]); ids[c - 1] = lo; ids[c] = hi; }
#pragma unroll
  for (int i = 0; i < kLeafMax; ++i) {
    if (i >= lf.count) continue;
    const int idx = ids[i];
    kmin = min(kmin, idx);
    const float4 pf = raw[idx];
    const double p[3] = {pf.x, pf.y, pf.z};
    for (int a = 0; a < 3; ++a) { b[a] += p[a]; for (int c = 0; c < 3; ++c) M[3 * a + c] += p[a] * p[c]; }
  }
  const int n = lf.count;
  const double mean[3] = {b[0] / n, b[1] / n, b[2] / n};
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kLeafMax; ++i) {
    if (i >= lf.count) continue;
    const float4 pf = raw[ids[i]];
    const double e[3] = {pf.x - mean[0], pf.y - mean[1], pf.z - mean[2]};
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] += e[a] * e[c];
  }
  leaf_id[l] = l;
  const int sc = scan_of_pos(ss, lf.start);
  kmin -= ss.offset[sc];                                    // index inside its own scan
  bool ok = n > 0 && rank3_sym(C) + 1 >= 3;               // :90-92
  double nv[3] = {0, 0, 0}, nn = 0;
  if (ok) {
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double invdet = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
    const double inv[9] = {c00 * invdet, (M[2] * M[7] - M[1] * M[8]) * invdet, (M[1] * M[5] - M[2] * M[4]) * invdet,
                           c01 * invdet, (M[0] * M[8] - M[2] * M[6]) * invdet, (M[2] * M[3] - M[0] * M[5]) * invdet,
                           c02 * invdet, (M[1] * M[6] - M[0] * M[7]) * invdet, (M[0] * M[4] - M[1] * M[3]) * invdet};
    for (int a = 0; a < 3; ++a) { nv[a] = inv[3 * a] * b[0] + inv[3 * a + 1] * b[1] + inv[3 * a + 2] * b[2]; nn += nv[a] * nv[a]; }   // :94
    nn = sqrt(nn);
    ok = nn > 0.0 && isfinite(nn);                         // singular M: dropped (the reference keeps a NaN normal)
  }
  if (ok) {
    leaf_p[l] = make_float4((float)mean[0], (float)mean[1], (float)mean[2], 0.f);
    leaf_n[l] = make_float4((float)(nv[0] / nn), (float)(nv[1] / nn), (float)(nv[2] / nn), 0.f);
    leaf_key[l] = ((unsigned long long)(uint32_t)sc << 32) | (uint32_t)kmin;
  } else {
    leaf_key[l] = ((unsigned long long)(uint32_t)sc << 32) | 0xffffffffull;     // sorts behind the scan's valid leaves
  }
}

// first sorted rank of every scan's leaves and the number of valid ones
__global__ void kd_scan_ranges(const unsigned long long* keys_sorted, const int32_t* counts, int S, int32_t* lstart, int32_t* m_out) {
  const int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc >= S) return;
  const int nl = counts[2];
  auto lower = [&](unsigned long long key) {       // first r with keys_sorted[r] >= key
    int lo = 0, hi = nl;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys_sorted[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int a = lower((unsigned long long)(uint32_t)sc << 32);
  const int b = lower(((unsigned long long)(uint32_t)sc << 32) | 0xffffffffull);
  lstart[sc] = a;
  m_out[sc] = b - a;
}

__global__ void kd_emit(const float4* leaf_p, const float4* leaf_n, const unsigned long long* keys_sorted, const int32_t* ids_sorted,
                        const int32_t* counts, const int32_t* lstart, const int32_t* out_offset, float4* out_p, float4* out_n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= counts[2]) return;
  const unsigned long long key = keys_sorted[r];
  if ((key & 0xffffffffull) == 0xffffffffull) return;      // dropped leaf
  const int sc = (int)(key >> 32);
  const int l = ids_sorted[r];
  const size_t o = (size_t)out_offset[sc] + (size_t)(r - lstart[sc]);
  out_p[o] = leaf_p[l];
  out_n[o] = leaf_n[l];
}

// ------------------------------------------------------------------------------------------
// The same forest with ONE workgroup per scan (kd_median_tree.h: radix select + partition per level, no sort): what a
// large batch uses.  A level of the sort-based formulation above is 7-8 full radix-sort passes over every (segment,
// coordinate) key of the batch -- 17 levels of them were two thirds of the sequence driver's wall time -- where
// cloud_types.cc:122-125 only asks for an nth_element.  One workgroup takes ~10 ms for a 120 k-point scan whatever the
// batch, so from a few dozen scans on (kForestMinScans) this form wins, and a batch of >= 256 scans fills the chip.
// Ties on a median value go left by the point's index in its scan (the sort-based form: by the previous level's order);
// both are legal outcomes of nth_element, and tie-free clouds give identical leaves.
// ------------------------------------------------------------------------------------------
constexpr int kForestMinScans = 32;

struct ForestDev {
  float4 *cur, *oth;             // [cap] working orders (point, .w = index in its scan)
  uint32_t *sid, *sid_o;         // [cap] segment of every position
  uint32_t *kk, *kk_o;           // [cap] the median select's keys (kd_median_tree.h)
  KdSeg* segs;                   // per scan: 2 x (n / 4 + 8) segments at seg_off[scan]
  uint2* nodes;                  // per scan: n / 2 + 8 nodes at node_off[scan]
  uint32_t* cnt;                 // per scan: 2 x (n / 4 + 8) fill counters at seg_off[scan]
  const int32_t* seg_off;        // [S]
  const int32_t* node_off;       // [S]
};

__global__ __launch_bounds__(kKdThreads) void kd_forest_build(const float4* raw, ScanSet ss, ForestDev f, int32_t* order, KdLeaf* leaves,
                                                              int32_t* counts, int32_t* status) {
  const int sc = blockIdx.x;
  const int n = ss.prefix[sc + 1] - ss.prefix[sc];
  if (n <= 0) return;
  const int tid = threadIdx.x;
  const int base = ss.prefix[sc];
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[kKdHistWords];   // (kd_median_build also lays 8-byte keys in it)
  __shared__ uint32_t s_w[17];
  __shared__ float s_box[6][16];
  __shared__ uint32_t s_misc[4];
  float4* cur = f.cur + base;
  float4* oth = f.oth + base;
  uint32_t* sid = f.sid + base;
  uint32_t* sid_o = f.sid_o + base;
  uint32_t* kk = f.kk + base;
  uint32_t* kk_o = f.kk_o + base;
  const int seg_cap = n / 4 + 8, node_cap = n / 2 + 8;
  KdSeg* seg = f.segs + f.seg_off[sc];
  KdSeg* seg_o = seg + seg_cap;
  uint2* nodes = f.nodes + f.node_off[sc];
  const float4* p0 = raw + ss.offset[sc];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += kKdThreads) {
    const float4 p = p0[i];
    cur[i] = make_float4(p.x, p.y, p.z, __int_as_float(i));
    sid[i] = 0;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if ((tid & 63) == 0) for (int d = 0; d < 3; ++d) { s_box[d][tid >> 6] = mn[d]; s_box[3 + d][tid >> 6] = mx[d]; }
  __syncthreads();
  if (tid == 0) {
    KdSeg r{};
    r.first = 0; r.count = (uint32_t)n; r.node = 0;
    for (int d = 0; d < 3; ++d) {
      float a = s_box[d][0], c = s_box[3 + d][0];
      for (int w = 1; w < kKdThreads / 64; ++w) { a = fminf(a, s_box[d][w]); c = fmaxf(c, s_box[3 + d][w]); }
      r.mn[d] = a; r.mx[d] = c;
    }
    seg[0] = r;
    s_misc[0] = 1;
  }
  __syncthreads();
  auto fetch = [&](uint32_t i) { const float4 p = p0[i]; return make_float4(p.x, p.y, p.z, __int_as_float((int)i)); };
  kd_median_build<kLeafMax, true>(n, fetch, cur, oth, sid, sid_o, seg, seg_o, kk, kk_o, nodes, f.cnt + f.seg_off[sc], seg_cap, node_cap, s_hist, s_w, s_misc, status);
  __syncthreads();
  // the leaves (position ranges of the forest) and the permutation: order[position] = index into raw
  const int nn = (int)s_misc[0];
  for (int v = tid; v < nn; v += kKdThreads) {
    const uint2 nd = nodes[v];
    if ((nd.y & 3u) != 3u) continue;
    const int slot = atomicAdd(&counts[2], 1);
    leaves[slot].start = base + (int)nd.x;
    leaves[slot].count = (int)(nd.y >> 2);
  }
  for (int i = tid; i < n; i += kKdThreads) order[base + i] = ss.offset[sc] + __float_as_int(cur[i].w);
}

}  // namespace

struct PrepWorkspace {
  int cap = 0;
  int32_t *order[2] = {nullptr, nullptr}, *seg[2] = {nullptr, nullptr}, *node_at[2] = {nullptr, nullptr};
  unsigned long long *keys[2] = {nullptr, nullptr};
  KdNode* nodes[2] = {nullptr, nullptr};
  KdLeaf* leaves = nullptr;
  int32_t* counts = nullptr;
  float* bbox = nullptr;
  float4 *leaf_p = nullptr, *leaf_n = nullptr;
  int32_t *leaf_id[2] = {nullptr, nullptr};
  void* sort_tmp = nullptr;
  size_t sort_bytes = 0;
  int32_t* m_dev = nullptr;              // [kMaxScans]
  int32_t* lstart = nullptr;             // [kMaxScans]
  int32_t* scan_meta = nullptr;          // device: offset[kMaxScans], prefix[kMaxScans + 1], out_offset[kMaxScans]
  int32_t* host_pinned = nullptr;        // [0..3] counts, then m[kMaxScans], then the scan_meta staging
  float4* avg_cent = nullptr;            // [cap] run centroids of prep_approx_voxel_grid (allocated on first use)
  int32_t* mb_meta = nullptr;            // prep_morton_sort_batch: device / pinned meta rows, and the event that guards the pinned ones
  int32_t* mb_host = nullptr;
  hipEvent_t mb_ev = nullptr;
  ForestDev forest{};                    // one-workgroup-per-scan forest (allocated on the first batch of >= kForestMinScans scans)
  int32_t* forest_meta = nullptr;        // device: seg_off[kMaxScans], node_off[kMaxScans]
  int32_t* forest_status = nullptr;
  bool forest_ready = false;
};


PrepWorkspace* prep_create(int max_points) {
  PrepWorkspace* w = new PrepWorkspace();
  w->cap = max_points;
  const size_t N = (size_t)max_points;
  bool ok = true;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; };
  for (int k = 0; k < 2; ++k) {
    A((void**)&w->order[k], N * 4); A((void**)&w->seg[k], N * 4); A((void**)&w->node_at[k], N * 4);
    A((void**)&w->keys[k], N * 8); A((void**)&w->nodes[k], (N / 4 + 16 + kMaxScans) * sizeof(KdNode)); A((void**)&w->leaf_id[k], (N / 2 + 16) * 4);
  }
  A((void**)&w->leaves, (N / 2 + 16) * sizeof(KdLeaf));
  A((void**)&w->counts, 16 * 4); A((void**)&w->bbox, (size_t)8 * 4 * kMaxScans); A((void**)&w->m_dev, (size_t)4 * kMaxScans);
  A((void**)&w->lstart, (size_t)4 * kMaxScans); A((void**)&w->scan_meta, (size_t)4 * (3 * kMaxScans + 8));
  A((void**)&w->leaf_p, (N / 2 + 16) * sizeof(float4)); A((void**)&w->leaf_n, (N / 2 + 16) * sizeof(float4));
  if (ok) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned)max_points, 0, 64, (hipStream_t)0);
    size_t bytes1 = 0;        // prep_sort_pairs' one-sweep configuration
    using OneSweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
    (void)rocprim::radix_sort_pairs<OneSweep>(nullptr, bytes1, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                              (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned)max_points, 0, 64, (hipStream_t)0);
    w->sort_bytes = std::max(bytes, bytes1) + 256;
    A(&w->sort_tmp, w->sort_bytes);
  }
  if (ok && hipHostMalloc((void**)&w->host_pinned, (size_t)4 * (8 + 6 * kMaxScans + 16)) != hipSuccess) ok = false;
  if (!ok) { prep_destroy(w); return nullptr; }
  return w;
}

void prep_destroy(PrepWorkspace* w) {
  if (!w) return;
  for (int k = 0; k < 2; ++k) {
    (void)hipFree(w->order[k]); (void)hipFree(w->seg[k]); (void)hipFree(w->node_at[k]); (void)hipFree(w->keys[k]);
    (void)hipFree(w->nodes[k]); (void)hipFree(w->leaf_id[k]);
  }
  (void)hipFree(w->leaves); (void)hipFree(w->counts); (void)hipFree(w->bbox); (void)hipFree(w->m_dev);
  (void)hipFree(w->lstart); (void)hipFree(w->scan_meta);
  (void)hipFree(w->leaf_p); (void)hipFree(w->leaf_n); (void)hipFree(w->sort_tmp); (void)hipFree(w->avg_cent);
  (void)hipFree(w->forest.cur); (void)hipFree(w->forest.oth); (void)hipFree(w->forest.sid); (void)hipFree(w->forest.sid_o); (void)hipFree(w->forest.kk); (void)hipFree(w->forest.kk_o);
  (void)hipFree(w->forest.segs); (void)hipFree(w->forest.nodes); (void)hipFree(w->forest.cnt); (void)hipFree(w->forest_meta); (void)hipFree(w->forest_status);
  if (w->host_pinned) (void)hipHostFree(w->host_pinned);
  (void)hipFree(w->mb_meta);
  if (w->mb_host) (void)hipHostFree(w->mb_host);
  if (w->mb_ev) (void)hipEventDestroy(w->mb_ev);
  delete w;
}

#define PCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

namespace {
__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__global__ void morton_keys(const float4* raw, int n, const float* bbox, unsigned long long* keys, int32_t* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = raw[i];
  const float inv = 1.0f / 0.0625f;                       // 6.25 cm quantum
  unsigned long long q[3];
  const float v[3] = {p.x, p.y, p.z};
  for (int c = 0; c < 3; ++c) {
    const float t = (v[c] - bbox[c]) * inv;
    q[c] = (isfinite(t)) ? (unsigned long long)fminf(fmaxf(t, 0.f), 2097151.f) : 0ull;
  }
  keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
  idx[i] = i;
}
__global__ void morton_gather(const float4* raw, const int32_t* idx, int n, float4* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
  float4 p = raw[i];
  p.w = __int_as_float(i);                                // the caller's index rides along
  out[k] = p;
}
// bbox over finite coordinates only (NaN points must not poison the quantisation origin): kFiniteMinBlocks partial minima,
// folded by a second one-wave launch (a single 1024-thread workgroup took 67 us per 120 k-point upload)
constexpr int kFiniteMinBlocks = 64;
__global__ __launch_bounds__(256) void finite_min_partial(const float4* raw, int n, float* part /*[kFiniteMinBlocks][3]*/) {
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = raw[i];
    if (isfinite(p.x)) mn[0] = fminf(mn[0], p.x);
    if (isfinite(p.y)) mn[1] = fminf(mn[1], p.y);
    if (isfinite(p.z)) mn[2] = fminf(mn[2], p.z);
  }
  __shared__ float s[4][3];
  for (int c = 0; c < 3; ++c)
    for (int off = 32; off > 0; off >>= 1) mn[c] = fminf(mn[c], __shfl_down(mn[c], off, 64));
  if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; ++c) s[threadIdx.x >> 6][c] = mn[c];
  __syncthreads();
  if (threadIdx.x < 3) part[3 * blockIdx.x + threadIdx.x] = fminf(fminf(s[0][threadIdx.x], s[1][threadIdx.x]), fminf(s[2][threadIdx.x], s[3][threadIdx.x]));
}
__global__ __launch_bounds__(64) void finite_min_fold(const float* part, float* out) {
  float v[3];
  for (int c = 0; c < 3; ++c) {
    v[c] = part[3 * threadIdx.x + c];                 // kFiniteMinBlocks == 64 lanes
    for (int off = 32; off > 0; off >>= 1) v[c] = fminf(v[c], __shfl_down(v[c], off, 64));
  }
  if (threadIdx.x == 0) for (int c = 0; c < 3; ++c) out[c] = isfinite(v[c]) ? v[c] : 0.f;
}
}  // namespace

unsigned long long* prep_keys(PrepWorkspace* w, int which) { return w->keys[which & 1]; }
int32_t* prep_values(PrepWorkspace* w, int which) { return w->order[which & 1]; }
hipError_t prep_sort_pairs(PrepWorkspace* w, hipStream_t st, int n, int end_bit) {
  if (!w || n <= 0 || n > w->cap) return hipErrorInvalidValue;
  size_t bytes = w->sort_bytes;
  // (rocPRIM sorts up to a million items by merging: ~18 launches for one 500 k-point target, against one histogram pass + one
  // launch per 8 key bits of its one-sweep radix sort -- and the caller passes as few bits as its keys can have)
  using OneSweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
  return rocprim::radix_sort_pairs<OneSweep>(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->order[1], (unsigned)n, 0, (unsigned)end_bit, st);
}

// Morton (Z-order) permutation of a cloud on the device: out[k] = raw[perm[k]], w = perm[k].
hipError_t prep_morton_sort(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float4* out) {
  if (!w || n <= 0 || n > w->cap) return hipErrorInvalidValue;
  const int gp = (n + 255) / 256;
  hipLaunchKernelGGL(finite_min_partial, dim3(kFiniteMinBlocks), dim3(256), 0, st, raw, n, w->bbox + 8);
  hipLaunchKernelGGL(finite_min_fold, dim3(1), dim3(64), 0, st, w->bbox + 8, w->bbox);
  hipLaunchKernelGGL(morton_keys, dim3(gp), dim3(256), 0, st, raw, n, w->bbox, w->keys[0], w->order[0]);
  size_t bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->order[1], (unsigned)n, 0, 63, st));
  hipLaunchKernelGGL(morton_gather, dim3(gp), dim3(256), 0, st, raw, w->order[1], n, out);
  return hipGetLastError();
}

// ---- S clouds at once -------------------------------------------------------------------------------------------
namespace {
struct MortonBatch {
  int32_t S;
  const int32_t* stage_off;      // [S] first row of cloud s in the staging array
  const int32_t* prefix;         // [S + 1] positions of the batch
  const long long* out_off;      // [S] first element of cloud s's output, relative to out_base
};
__global__ __launch_bounds__(1024) void finite_min_batch(const float4* stage, MortonBatch mb, float* bbox /*[S][4]*/) {
  const int sc = blockIdx.x;
  const float4* p0 = stage + mb.stage_off[sc];
  const int n = mb.prefix[sc + 1] - mb.prefix[sc];
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float4 p = p0[i];
    if (isfinite(p.x)) mn[0] = fminf(mn[0], p.x);
    if (isfinite(p.y)) mn[1] = fminf(mn[1], p.y);
    if (isfinite(p.z)) mn[2] = fminf(mn[2], p.z);
  }
  __shared__ float s[16][3];
  for (int c = 0; c < 3; ++c)
    for (int off = 32; off > 0; off >>= 1) mn[c] = fminf(mn[c], __shfl_down(mn[c], off, 64));
  if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; ++c) s[threadIdx.x >> 6][c] = mn[c];
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < 16; ++w) v = fminf(v, s[w][threadIdx.x]);
    bbox[4 * sc + threadIdx.x] = isfinite(v) ? v : 0.f;
  }
}
__device__ __forceinline__ int morton_batch_cloud(const MortonBatch& mb, int pos) {
  int lo = 0, hi = mb.S - 1;                       // last s with prefix[s] <= pos
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (mb.prefix[mid] <= pos) lo = mid; else hi = mid - 1; }
  return lo;
}
__global__ void morton_keys_batch(const float4* stage, MortonBatch mb, int total, const float* bbox, unsigned long long* keys, int32_t* idx) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= total) return;
  const int sc = morton_batch_cloud(mb, pos);
  const float4 p = stage[mb.stage_off[sc] + (pos - mb.prefix[sc])];
  const float inv = 1.0f / 0.0625f;                       // the quantum of morton_keys
  unsigned long long q[3];
  const float v[3] = {p.x, p.y, p.z};
  for (int c = 0; c < 3; ++c) {
    const float t = (v[c] - bbox[4 * sc + c]) * inv;
    q[c] = (isfinite(t)) ? (unsigned long long)fminf(fmaxf(t, 0.f), 262143.f) : 0ull;     // 18 bits per axis
  }
  keys[pos] = ((unsigned long long)sc << 54) | spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
  idx[pos] = pos;
}
__global__ void morton_gather_batch(const float4* stage, MortonBatch mb, int total, const unsigned long long* keys_sorted, const int32_t* idx_sorted,
                                    float4* out_base) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  const int sc = (int)(keys_sorted[k] >> 54);                 // clouds are contiguous in cloud order after the sort
  const int i = idx_sorted[k] - mb.prefix[sc];                // the row's index in its own cloud
  float4 p = stage[mb.stage_off[sc] + i];
  p.w = __int_as_float(i);                                    // the caller's index rides along
  out_base[mb.out_off[sc] + (k - mb.prefix[sc])] = p;
}
}  // namespace

hipError_t prep_morton_sort_batch(PrepWorkspace* w, hipStream_t st, const float4* stage, int S, const int* stage_off, const int* n,
                                  const long long* out_off, float4* out_base) {
  if (!w || S <= 0 || S > kMaxScans) return hipErrorInvalidValue;
  if (!w->mb_meta) {
    if (hipMalloc((void**)&w->mb_meta, sizeof(int32_t) * (2 * kMaxScans + 2) + sizeof(long long) * kMaxScans + 16) != hipSuccess) return hipErrorOutOfMemory;
    if (hipHostMalloc((void**)&w->mb_host, sizeof(int32_t) * (2 * kMaxScans + 2) + sizeof(long long) * kMaxScans + 16) != hipSuccess) return hipErrorOutOfMemory;
  }
  // the staging of the meta rows is re-used by the next call: the previous copy must have left the host
  if (w->mb_ev) { PCHK(hipEventSynchronize(w->mb_ev)); } else { PCHK(hipEventCreateWithFlags(&w->mb_ev, hipEventDisableTiming)); }
  int32_t* h_off = w->mb_host;
  int32_t* h_pre = w->mb_host + kMaxScans;
  long long* h_out = reinterpret_cast<long long*>(w->mb_host + 2 * kMaxScans + 2);
  long long total = 0;
  for (int s2 = 0; s2 < S; ++s2) {
    if (n[s2] <= 0) return hipErrorInvalidValue;
    h_off[s2] = stage_off[s2]; h_pre[s2] = (int32_t)total; h_out[s2] = out_off[s2];
    total += n[s2];
  }
  h_pre[S] = (int32_t)total;
  if (total > w->cap) return hipErrorInvalidValue;
  const size_t meta_bytes = sizeof(int32_t) * (2 * kMaxScans + 2) + sizeof(long long) * kMaxScans;
  PCHK(hipMemcpyAsync(w->mb_meta, w->mb_host, meta_bytes, hipMemcpyHostToDevice, st));
  PCHK(hipEventRecord(w->mb_ev, st));
  MortonBatch mb;
  mb.S = S; mb.stage_off = w->mb_meta; mb.prefix = w->mb_meta + kMaxScans;
  mb.out_off = reinterpret_cast<const long long*>(w->mb_meta + 2 * kMaxScans + 2);
  const int N = (int)total, gp = (N + 255) / 256;
  hipLaunchKernelGGL(finite_min_batch, dim3(S), dim3(1024), 0, st, stage, mb, w->bbox);
  hipLaunchKernelGGL(morton_keys_batch, dim3(gp), dim3(256), 0, st, stage, mb, N, w->bbox, w->keys[0], w->order[0]);
  int id_bits = 1;
  while ((1 << id_bits) < S) ++id_bits;
  size_t bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->order[1], (unsigned)N, 0, (unsigned)(54 + id_bits), st));
  hipLaunchKernelGGL(morton_gather_batch, dim3(gp), dim3(256), 0, st, stage, mb, N, w->keys[1], w->order[1], out_base);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// libpointmatcher's RandomSamplingDataPointsFilter (reading filter of the IcpUsingPointMatcher chain,
// /root/reference/registrators/icp_pointmatcher.cc:170-174) on a Morton-ordered device cloud whose .w holds the caller
// index.  The reference draws rand() per point; here a point is kept when its counter-based uniform -- a pure function
// of (seed, CALLER index), the generator of the front-end RandomSampler in cloud_filters.hip -- is < prob, so the kept
// set is reproducible on the host (staticmapping_amd.matcher.sampling_mask).  Output: the kept points in the input's
// (Morton) order with .w = the point's index in the sampled cloud in CALLER order.  Two exclusive scans.
// ------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ double sample_uniform(uint32_t seed, uint32_t i) {
  unsigned long long z = ((unsigned long long)seed << 32 | i) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__global__ void sample_flags(const float4* in, int n, uint32_t seed, float prob, int32_t* by_caller, int32_t* by_pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  by_caller[i] = (prob >= 1.0f || sample_uniform(seed, (uint32_t)i) < (double)prob) ? 1 : 0;
  const uint32_t c = (uint32_t)__float_as_int(in[i].w);
  by_pos[i] = (prob >= 1.0f || sample_uniform(seed, c) < (double)prob) ? 1 : 0;
}
__global__ void sample_scatter(const float4* in, int n, const int32_t* by_pos, const int32_t* pos_rank, const int32_t* caller_rank,
                               float4* out, int32_t* count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (by_pos[i]) {
    float4 p = in[i];
    p.w = __int_as_float(caller_rank[__float_as_int(p.w)]);
    out[pos_rank[i]] = p;
  }
  if (i == n - 1) count[0] = pos_rank[i] + by_pos[i];
}
}  // namespace

hipError_t prep_sample_morton(PrepWorkspace* w, hipStream_t st, const float4* in, int n, float prob, uint32_t seed, float4* out, int* m_host) {
  if (!w || n <= 0 || n > w->cap || in == out) return hipErrorInvalidValue;
  const int gp = (n + 255) / 256;
  int32_t *fc = w->order[0], *fp = w->order[1], *rc = w->seg[0], *rp = w->seg[1];
  hipLaunchKernelGGL(sample_flags, dim3(gp), dim3(256), 0, st, in, n, seed, prob, fc, fp);
  size_t bytes = w->sort_bytes;
  PCHK(rocprim::exclusive_scan(w->sort_tmp, bytes, fc, rc, 0, (size_t)n, rocprim::plus<int32_t>(), st));
  bytes = w->sort_bytes;
  PCHK(rocprim::exclusive_scan(w->sort_tmp, bytes, fp, rp, 0, (size_t)n, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(sample_scatter, dim3(gp), dim3(256), 0, st, in, n, fp, rp, rc, out, w->counts);
  PCHK(hipMemcpyAsync(w->host_pinned, w->counts, 4, hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  *m_host = w->host_pinned[0];
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// pcl::ApproximateVoxelGrid<PointXYZ>::applyFilter (PCL 1.8.1, pcl/filters/impl/approximate_voxel_grid.hpp), the
// down-sampling step of registrators/ndt_gicp.cc:60-71.  The reference loop is serial: a 512-entry hash history
// indexed by (ix * 7171 + iy * 3079 + iz * 4231) & 511; a point that lands on an entry holding a DIFFERENT voxel
// flushes that entry's float centroid to the output, and whatever is left is flushed at the end in entry order.
// Each entry only ever sees its own subsequence of the points, so the filter decomposes exactly:
//   sort by (entry, arrival index); maximal runs of equal voxel inside an entry are the flushed centroids (summed
//   in arrival order, in float, like the reference); a run is emitted when the first point of the entry's next run
//   arrives, the last run of every entry at time n + entry.  A second sort by that time gives the output order.
// ------------------------------------------------------------------------------------------
namespace {
constexpr int kAvgHist = 512;
__device__ __forceinline__ void avg_voxel(const float4 p, float inv, int& ix, int& iy, int& iz, uint32_t& hsh) {
  ix = (int)floorf(p.x * inv); iy = (int)floorf(p.y * inv); iz = (int)floorf(p.z * inv);
  hsh = (uint32_t)((ix * 7171 + iy * 3079 + iz * 4231) & (kAvgHist - 1));
}
__global__ void avg_keys(const float4* raw, int n, float inv, unsigned long long* keys, int32_t* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int ix, iy, iz; uint32_t hsh;
  avg_voxel(raw[i], inv, ix, iy, iz, hsh);
  keys[i] = ((unsigned long long)hsh << 32) | (uint32_t)i;
  idx[i] = i;
}
__global__ void avg_heads(const float4* raw, const int32_t* idx, int n, float inv, int32_t* head) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int ax, ay, az, bx, by, bz; uint32_t ha, hb;
  avg_voxel(raw[idx[s]], inv, ax, ay, az, ha);
  int hd = 1;
  if (s > 0) {
    avg_voxel(raw[idx[s - 1]], inv, bx, by, bz, hb);
    hd = (ha != hb || ax != bx || ay != by || az != bz) ? 1 : 0;
  }
  head[s] = hd;
}
__global__ void avg_run_starts(const int32_t* head, const int32_t* incl, int n, int32_t* run_start, int32_t* counts) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (head[s]) run_start[incl[s] - 1] = s;
  if (s == n - 1) counts[0] = incl[s];
}
__global__ void avg_flush(const float4* raw, const int32_t* idx, const int32_t* run_start, const int32_t* counts, int n, float inv,
                          float4* cent, unsigned long long* when, int32_t* rid) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = counts[0];
  if (r >= R) return;
  const int a = run_start[r], b = (r + 1 < R) ? run_start[r + 1] : n;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int s = a; s < b; ++s) { const float4 p = raw[idx[s]]; sx += p.x; sy += p.y; sz += p.z; }   // hhe->centroid += scratch
  const float c = (float)(b - a);
  cent[r] = make_float4(sx / c, sy / c, sz / c, 0.f);                                               // flush: centroid /= count
  int ix, iy, iz; uint32_t h0, h1 = 0xffffffffu;
  avg_voxel(raw[idx[a]], inv, ix, iy, iz, h0);
  if (b < n) avg_voxel(raw[idx[b]], inv, ix, iy, iz, h1);
  when[r] = (h1 == h0) ? (unsigned long long)(uint32_t)idx[b] : (unsigned long long)n + h0;
  rid[r] = r;
}
__global__ void avg_emit(const float4* cent, const int32_t* rid_sorted, const int32_t* counts, float4* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= counts[0]) return;
  float4 c = cent[rid_sorted[k]];
  c.w = __int_as_float(k);
  out[k] = c;
}
}  // namespace

hipError_t prep_approx_voxel_grid(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float leaf, float4* out, int* m_host) {
  if (!w || n <= 0 || n > w->cap || !(leaf > 0.f)) return hipErrorInvalidValue;
  if (!w->avg_cent && hipMalloc((void**)&w->avg_cent, sizeof(float4) * (size_t)w->cap) != hipSuccess) return hipErrorOutOfMemory;
  const float inv = 1.0f / leaf;                             // inverse_leaf_size_ = Array3f::Ones() / leaf_size_
  const int gp = (n + 255) / 256;
  hipLaunchKernelGGL(avg_keys, dim3(gp), dim3(256), 0, st, raw, n, inv, w->keys[0], w->order[0]);
  size_t bytes = w->sort_bytes;
  // (the pairs start in arrival order and the sort is stable: the nine bits of the history entry are all it has to look at)
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->order[1], (unsigned)n, 32, 41, st));
  hipLaunchKernelGGL(avg_heads, dim3(gp), dim3(256), 0, st, raw, w->order[1], n, inv, w->seg[0]);
  size_t sbytes = w->sort_bytes;
  PCHK(rocprim::inclusive_scan(w->sort_tmp, sbytes, w->seg[0], w->seg[1], (size_t)n, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(avg_run_starts, dim3(gp), dim3(256), 0, st, w->seg[0], w->seg[1], n, w->node_at[0], w->counts);
  hipLaunchKernelGGL(avg_flush, dim3(gp), dim3(256), 0, st, raw, w->order[1], w->node_at[0], w->counts, n, inv, w->avg_cent,
                     w->keys[0], w->order[0]);
  PCHK(hipMemcpyAsync(w->host_pinned, w->counts, 4, hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  const int R = w->host_pinned[0];
  if (R <= 0 || R > n) return hipErrorUnknown;
  bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->node_at[1], (unsigned)R, 0, 33, st));
  hipLaunchKernelGGL(avg_emit, dim3((R + 255) / 256), dim3(256), 0, st, w->avg_cent, w->node_at[1], w->counts, out);
  *m_host = R;
  return hipGetLastError();
}

// ---- S clouds at once: one key pass, ONE stable sort on (cloud, history entry), one scan, one flush, one sort of the runs by
// (cloud, emission time), one emit; the counts come back with a single synchronise.  Per cloud the output is exactly
// prep_approx_voxel_grid's.  The workspace must hold sum(n) points; S <= kAvgMaxClouds.
namespace {
constexpr int kAvgMaxClouds = 64;
struct AvgClouds {
  int32_t S;
  int32_t prefix[kAvgMaxClouds + 1];         // positions of the clouds in the batch
  const float4* raw[kAvgMaxClouds];
  float4* out[kAvgMaxClouds];
};
__device__ __forceinline__ int avg_cloud_of(const AvgClouds& cl, int g) {
  int c = 0;
  while (c + 1 < cl.S && g >= cl.prefix[c + 1]) ++c;
  return c;
}
__global__ void avg_keys_b(const AvgClouds cl, int N, float inv, unsigned long long* keys, int32_t* idx) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N) return;
  const int c = avg_cloud_of(cl, g);
  int ix, iy, iz; uint32_t hsh;
  avg_voxel(cl.raw[c][g - cl.prefix[c]], inv, ix, iy, iz, hsh);
  keys[g] = ((unsigned long long)c << 9) | hsh;
  idx[g] = g;
}
__global__ void avg_heads_b(const AvgClouds cl, const int32_t* idx, int N, float inv, int32_t* head) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int g = idx[s], c = avg_cloud_of(cl, g);
  int ax, ay, az, bx, by, bz; uint32_t ha, hb;
  avg_voxel(cl.raw[c][g - cl.prefix[c]], inv, ax, ay, az, ha);
  int hd = 1;
  if (s > cl.prefix[c]) {                                     // (a cloud's points fill the sorted positions [prefix[c], prefix[c + 1]))
    const int gp = idx[s - 1];
    avg_voxel(cl.raw[c][gp - cl.prefix[c]], inv, bx, by, bz, hb);
    hd = (ha != hb || ax != bx || ay != by || az != bz) ? 1 : 0;
  }
  head[s] = hd;
}
__global__ void avg_run_starts_b(const AvgClouds cl, const int32_t* head, const int32_t* incl, int N, int32_t* run_start, int32_t* runp) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  if (head[s]) run_start[incl[s] - 1] = s;
  if (s <= cl.S) runp[s] = s < cl.S ? incl[cl.prefix[s]] - 1 : incl[N - 1];      // first run of every cloud, and the total
}
__global__ void avg_flush_b(const AvgClouds cl, const int32_t* idx, const int32_t* run_start, const int32_t* runp, int N, float inv,
                            float4* cent, unsigned long long* when, int32_t* rid) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = runp[cl.S];
  if (r >= R) return;
  const int a = run_start[r], b = (r + 1 < R) ? run_start[r + 1] : N;
  const int c = avg_cloud_of(cl, idx[a]);
  const float4* raw = cl.raw[c];
  const int base = cl.prefix[c], nc = cl.prefix[c + 1] - base;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int s = a; s < b; ++s) { const float4 p = raw[idx[s] - base]; sx += p.x; sy += p.y; sz += p.z; }   // hhe->centroid += scratch
  const float cnt = (float)(b - a);
  cent[r] = make_float4(sx / cnt, sy / cnt, sz / cnt, 0.f);                                           // flush: centroid /= count
  int ix, iy, iz; uint32_t h0, h1 = 0xffffffffu;
  avg_voxel(raw[idx[a] - base], inv, ix, iy, iz, h0);
  if (b < cl.prefix[c + 1]) avg_voxel(raw[idx[b] - base], inv, ix, iy, iz, h1);
  const unsigned long long w = (h1 == h0) ? (unsigned long long)(uint32_t)(idx[b] - base) : (unsigned long long)nc + h0;
  when[r] = ((unsigned long long)c << 33) | w;
  rid[r] = r;
}
__global__ void avg_emit_b(const AvgClouds cl, const float4* cent, const int32_t* rid_sorted, const int32_t* runp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= runp[cl.S]) return;
  int c = 0;
  while (c + 1 < cl.S && k >= runp[c + 1]) ++c;
  float4 v = cent[rid_sorted[k]];
  v.w = __int_as_float(k - runp[c]);
  cl.out[c][k - runp[c]] = v;
}
}  // namespace

hipError_t prep_approx_voxel_grid_batch(PrepWorkspace* w, hipStream_t st, int S, const float4* const* raw, const int* n, float leaf, float4* const* out, int* m_host) {
  if (!w || S <= 0 || S > kAvgMaxClouds || !(leaf > 0.f)) return hipErrorInvalidValue;
  AvgClouds cl{};
  cl.S = S;
  long long total = 0;
  for (int c = 0; c < S; ++c) {
    if (n[c] <= 0) return hipErrorInvalidValue;
    cl.prefix[c] = (int32_t)total; cl.raw[c] = raw[c]; cl.out[c] = out[c];
    total += n[c];
  }
  cl.prefix[S] = (int32_t)total;
  if (total > w->cap) return hipErrorInvalidValue;
  if (!w->avg_cent && hipMalloc((void**)&w->avg_cent, sizeof(float4) * (size_t)w->cap) != hipSuccess) return hipErrorOutOfMemory;
  const float inv = 1.0f / leaf;
  const int N = (int)total, gp = (N + 255) / 256;
  int cbits = 0;
  while ((1 << cbits) < S) ++cbits;
  hipLaunchKernelGGL(avg_keys_b, dim3(gp), dim3(256), 0, st, cl, N, inv, w->keys[0], w->order[0]);
  size_t bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->order[1], (unsigned)N, 0, (unsigned)(9 + cbits), st));
  hipLaunchKernelGGL(avg_heads_b, dim3(gp), dim3(256), 0, st, cl, w->order[1], N, inv, w->seg[0]);
  size_t sbytes = w->sort_bytes;
  PCHK(rocprim::inclusive_scan(w->sort_tmp, sbytes, w->seg[0], w->seg[1], (size_t)N, rocprim::plus<int32_t>(), st));
  int32_t* runp = w->scan_meta;                               // [S + 1] (kAvgMaxClouds + 1 <= 3 * kMaxScans + 8)
  hipLaunchKernelGGL(avg_run_starts_b, dim3(gp), dim3(256), 0, st, cl, w->seg[0], w->seg[1], N, w->node_at[0], runp);
  hipLaunchKernelGGL(avg_flush_b, dim3(gp), dim3(256), 0, st, cl, w->order[1], w->node_at[0], runp, N, inv, w->avg_cent, w->keys[0], w->order[0]);
  PCHK(hipMemcpyAsync(w->host_pinned, runp, sizeof(int32_t) * (S + 1), hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  const int R = w->host_pinned[S];
  if (R <= 0 || R > N) return hipErrorUnknown;
  bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[0], w->node_at[1], (unsigned)R, 0, (unsigned)(33 + cbits), st));
  hipLaunchKernelGGL(avg_emit_b, dim3((R + 255) / 256), dim3(256), 0, st, cl, w->avg_cent, w->node_at[1], runp);
  for (int c = 0; c < S; ++c) m_host[c] = w->host_pinned[c + 1] - w->host_pinned[c];
  return hipGetLastError();
}

// the arrays of the one-workgroup-per-scan forest (batches of >= kForestMinScans scans), allocated on first use or ahead of it
hipError_t prep_reserve_forest(PrepWorkspace* w) {
  if (!w) return hipErrorInvalidValue;
  if (w->forest_ready) return hipSuccess;
  const size_t C = (size_t)w->cap, SG = 2 * (C / 4 + 8 * (size_t)kMaxScans), ND = C / 2 + 8 * (size_t)kMaxScans;
  const bool ok = hipMalloc((void**)&w->forest.cur, C * 16) == hipSuccess && hipMalloc((void**)&w->forest.oth, C * 16) == hipSuccess &&
                  hipMalloc((void**)&w->forest.sid, C * 4) == hipSuccess && hipMalloc((void**)&w->forest.sid_o, C * 4) == hipSuccess &&
                  hipMalloc((void**)&w->forest.kk, C * 4) == hipSuccess && hipMalloc((void**)&w->forest.kk_o, C * 4) == hipSuccess &&
                  hipMalloc((void**)&w->forest.segs, SG * sizeof(KdSeg)) == hipSuccess && hipMalloc((void**)&w->forest.nodes, ND * sizeof(uint2)) == hipSuccess &&
                  hipMalloc((void**)&w->forest.cnt, SG * 4) == hipSuccess && hipMalloc((void**)&w->forest_meta, sizeof(int32_t) * 2 * kMaxScans) == hipSuccess &&
                  hipMalloc((void**)&w->forest_status, 4) == hipSuccess;
  if (!ok) return hipErrorOutOfMemory;
  w->forest.seg_off = w->forest_meta; w->forest.node_off = w->forest_meta + kMaxScans;
  w->forest_ready = true;
  return hipSuccess;
}

// S scans at once: scan s = raw[offset[s] .. offset[s] + n[s]), results to out_p/out_n[out_offset[s] ..], m_host[s] survivors.
hipError_t prep_calculate_normals_batch(PrepWorkspace* w, hipStream_t st, const float4* raw, int S, const int* offset, const int* n,
                                        const int* out_offset, float4* out_p, float4* out_n, int* m_host) {
  if (!w || S <= 0 || S > kMaxScans) return hipErrorInvalidValue;
  int32_t* hp = w->host_pinned;
  int32_t* h_meta = hp + 8 + kMaxScans;           // offset[S], prefix[S + 1], out_offset[S]
  long long total = 0;
  int nmax = 0;
  for (int s2 = 0; s2 < S; ++s2) {
    if (n[s2] <= 0) return hipErrorInvalidValue;
    h_meta[s2] = offset[s2];
    h_meta[kMaxScans + s2] = (int32_t)total;
    h_meta[2 * kMaxScans + 1 + s2] = out_offset[s2];
    total += n[s2];
    nmax = std::max(nmax, n[s2]);
  }
  h_meta[kMaxScans + S] = (int32_t)total;
  if (total > w->cap) return hipErrorInvalidValue;
  PCHK(hipMemcpyAsync(w->scan_meta, h_meta, sizeof(int32_t) * (3 * kMaxScans + 2), hipMemcpyHostToDevice, st));
  ScanSet ss;
  ss.S = S; ss.offset = w->scan_meta; ss.prefix = w->scan_meta + kMaxScans;
  const int32_t* d_out_offset = w->scan_meta + 2 * kMaxScans + 1;
  const int N = (int)total;
  const int gp = (N + 255) / 256;
  int cur = 0;
  const bool use_forest = S >= kForestMinScans;
  if (use_forest) {
    // one workgroup per scan, no sorts: the leaves and the permutation come out of one launch
    PCHK(prep_reserve_forest(w));
    int32_t* h_f = h_meta + 3 * kMaxScans + 2;           // staging behind the scan meta
    long long so = 0, no = 0;
    for (int s2 = 0; s2 < S; ++s2) {
      h_f[s2] = (int32_t)so; h_f[kMaxScans + s2] = (int32_t)no;
      so += 2ll * (n[s2] / 4 + 8); no += n[s2] / 2 + 8;
    }
    PCHK(hipMemcpyAsync(w->forest_meta, h_f, sizeof(int32_t) * 2 * kMaxScans, hipMemcpyHostToDevice, st));
    PCHK(hipMemsetAsync(w->counts, 0, 16, st));
    PCHK(hipMemsetAsync(w->forest_status, 0, 4, st));
    hipLaunchKernelGGL(kd_forest_build, dim3(S), dim3(kKdThreads), 0, st, raw, ss, w->forest, w->order[0], w->leaves, w->counts, w->forest_status);
  } else {
  hipLaunchKernelGGL(kd_bbox, dim3(S), dim3(1024), 0, st, raw, ss, w->bbox);
  hipLaunchKernelGGL(kd_init, dim3(gp), dim3(256), 0, st, ss, N, w->order[0], w->seg[0], w->node_at[0], w->nodes[0], w->counts, w->bbox, w->leaves);
  hipLaunchKernelGGL(kd_roots, dim3((S + 63) / 64), dim3(64), 0, st, ss, w->node_at[0], w->nodes[0], w->counts, w->bbox, w->leaves);
  // depth <= ceil(log2(n / 4)) + 1; a fixed number of levels runs (extra levels are no-ops: zero active nodes)
  int levels = 1;
  while (((long long)kLeafMax << levels) < (long long)nmax * 2) ++levels;
  levels += 1;
  int seg_bits = 1;
  while ((1ll << seg_bits) < total) ++seg_bits;
  for (int lv = 0; lv < levels; ++lv) {
    const int nxt = cur ^ 1;
    const long long cap_nodes = std::min<long long>(total / (kLeafMax + 1) + 2 + S, (long long)S << std::min(lv, 24));
    const int gn = (int)((cap_nodes + 63) / 64);
    hipLaunchKernelGGL(kd_choose_dim, dim3(gn), dim3(64), 0, st, w->nodes[cur], w->counts);
    hipLaunchKernelGGL(kd_keys, dim3(gp), dim3(256), 0, st, raw, N, w->order[cur], w->seg[cur], w->node_at[cur], w->nodes[cur], w->keys[0]);
    size_t bytes = w->sort_bytes;
    // key = segment start (needs `seg_bits` bits) << 32 | coordinate bits
    PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->order[cur], w->order[nxt], (unsigned)N, 0, 32 + seg_bits, st));
    PCHK(hipMemsetAsync(w->node_at[nxt], 0xff, (size_t)N * 4, st));
    hipLaunchKernelGGL(kd_split, dim3(gn), dim3(64), 0, st, raw, w->order[nxt], w->nodes[cur], w->nodes[nxt], w->counts, w->node_at[nxt], w->leaves);
    hipLaunchKernelGGL(kd_update_seg, dim3(gp), dim3(256), 0, st, N, w->seg[cur], w->node_at[cur], w->nodes[cur], w->seg[nxt]);
    hipLaunchKernelGGL(kd_advance, dim3(1), dim3(1), 0, st, w->counts);
    cur = nxt;
  }
  }
  // leaves -> (mean, normal), ordered per scan by the smallest original index of the leaf (cloud_types.cc:358)
  const int max_leaves = N / 2 + 16;
  const int gl = (max_leaves + 255) / 256;
  hipLaunchKernelGGL(kd_leaf_normals, dim3(gl), dim3(256), 0, st, raw, ss, w->order[cur], w->leaves, w->counts, w->leaf_p, w->leaf_n, w->keys[0], w->leaf_id[0]);
  PCHK(hipMemcpyAsync(hp, w->counts, 12, hipMemcpyDeviceToHost, st));
  const bool forest_used = S >= kForestMinScans;
  if (forest_used) PCHK(hipMemcpyAsync(hp + 3, w->forest_status, 4, hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  const int nl = hp[2];
  for (int s2 = 0; s2 < S; ++s2) m_host[s2] = 0;
  // kd_median_build reports a capacity overflow of its node / segment arrays here (sized for the leaf-size bounds of the split
  // rule, so this means corrupted input): an incomplete tree must not be handed on as a prepared target
  if (forest_used && hp[3] != 0) return hipErrorInvalidValue;
  if (nl <= 0 || nl > max_leaves) return hipSuccess;
  size_t bytes = w->sort_bytes;
  PCHK(rocprim::radix_sort_pairs(w->sort_tmp, bytes, w->keys[0], w->keys[1], w->leaf_id[0], w->leaf_id[1], (unsigned)nl, 0, 64, st));
  hipLaunchKernelGGL(kd_scan_ranges, dim3((S + 63) / 64), dim3(64), 0, st, w->keys[1], w->counts, S, w->lstart, w->m_dev);
  hipLaunchKernelGGL(kd_emit, dim3((nl + 255) / 256), dim3(256), 0, st, w->leaf_p, w->leaf_n, w->keys[1], w->leaf_id[1], w->counts, w->lstart,
                     d_out_offset, out_p, out_n);
  PCHK(hipMemcpyAsync(hp + 8, w->m_dev, sizeof(int32_t) * S, hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  PCHK(hipGetLastError());
  for (int s2 = 0; s2 < S; ++s2) m_host[s2] = hp[8 + s2];
  return hipSuccess;
}

hipError_t prep_calculate_normals(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float4* out_p, float4* out_n, int* m_host) {
  const int zero = 0;
  return prep_calculate_normals_batch(w, st, raw, 1, &zero, &n, &zero, out_p, out_n, m_host);
}

}  // namespace smhip
