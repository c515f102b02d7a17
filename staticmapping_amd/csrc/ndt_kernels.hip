// ndt_kernels.hip -- hand-written gfx950 kernels for registrators::Ndt (pclomp NDT).
//
// Reference lines are in /root/reference/registrators:
//   ndt_voxel_*      VoxelGridCovariance::applyFilter   pclomp/voxel_grid_covariance_omp_impl.hpp:49-370
//   ndt_derivatives  computeDerivatives + computePointDerivatives + updateDerivatives
//                                                       pclomp/ndt_omp_impl.hpp:180-284, 397-438, 483-535
//   ndt_reduce       the per-thread partial fold        pclomp/ndt_omp_impl.hpp:272-281
// The Newton / More-Thuente driver (ndt_omp_impl.hpp:81-171, 757-916) is 6-vector host code in
// smhip_ndt_api.hip.  As in the reference the per-neighbour math is float and the accumulation double.
#pragma once
#include "smhip_device.h"

namespace smhip {

constexpr int kNdtMaxWords = 1 << 20;        // 32 voxels per word: up to 32 Mi voxels in the dense box
constexpr int kNdtDerivThreads = 256;
constexpr int kNdtDerivCols = 44;            // score + 6 gradient + 36 hessian + pair count
constexpr int kNdtMaxDerivBlocks = 2048;

struct NdtVoxel {          // 64 B record of one searchable voxel (n >= min_points)
  double mean[3];          // Leaf::mean_  (double)
  float icov[6];           // Leaf::icov_ cast to float: xx xy xz yy yz zz
  float centroid[3];       // Leaf::centroid (float)
  int32_t n;               // nr_points; -1 = eigen check failed (stays searchable with icov = 0)
};

struct NdtGridInfo {       // written by ndt_voxel_setup
  int32_t min_b[3];
  int32_t div_b[3];
  int32_t wx;              // words per x row
  int32_t nw;              // total words
  int32_t nocc;            // occupied voxels
  int32_t status;          // 0 ok, 7 = voxel box larger than the bit grid
  float inv;               // inverse leaf size (float, as PCL)
  float res;
};

// One voxel table (one pair slot's target).  The kernels take a device ARRAY of these and pick theirs by block index: a batch
// of K Aligns builds its K tables and evaluates its K derivative sets in single launches (grid.y = pair).
struct NdtDev {
  int32_t nt, ns;
  int32_t min_points;
  float eig_mult;
  const float4* tgt;       // raw target points
  const float4* src;       // raw source points (Morton order)
  NdtGridInfo* info;
  const double* tpart;     // [kTgtReduceBlocks][16] from tgt_reduce
  uint32_t* bits;          // [kNdtMaxWords]
  uint2* words;            // [kNdtMaxWords]
  uint32_t* vstart;        // [nt + 1]
  float4* vpts;            // [nt] points sorted by voxel
  NdtVoxel* vox;           // [nt] one record per occupied voxel
  double* icovd;           // [nt][6] Leaf::icov_ in double: xx xy xz yy yz zz (stock PCL path reads these)
  double* partials;        // [kNdtMaxDerivBlocks][kNdtDerivCols]
  double* out;             // [kNdtDerivCols]
  int32_t key_off;         // where this table's (voxel code, point) pairs start in the batch's sort arrays
  int32_t pad_;
};

struct NdtPose {           // per derivative evaluation
  float T[12];             // final_transformation_ rows 0..2 (float 4x4)
  float j_ang[8][3];       // computeAngleDerivatives (float rows)
  float h_ang[15][3];
  float d1, d2;            // gauss_d1_, gauss_d2_ (d1 kept in double on the host too, see use)
  double d1d, d2d;
  double j_angd[8][3];     // the same rows in double (stock PCL path)
  double h_angd[15][3];
  float res2;              // resolution^2 for the centroid radius test
  int32_t compute_hessian;
};

// ------------------------------------------------------------------------------------------
// voxel grid build
// ------------------------------------------------------------------------------------------
__global__ void ndt_voxel_setup(const NdtDev* __restrict__ devs, float leaf) {
  if (threadIdx.x != 0) return;
  const NdtDev d = devs[blockIdx.x];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < kTgtReduceBlocks; ++k)
    for (int c = 0; c < 3; ++c) {
      mn[c] = fminf(mn[c], (float)d.tpart[16 * k + 3 + c]);
      mx[c] = fmaxf(mx[c], (float)d.tpart[16 * k + 6 + c]);
    }
  NdtGridInfo* g = d.info;
  const float inv = 1.0f / leaf;                                   // inverse_leaf_size_
  g->inv = inv; g->res = leaf;
  double vox = 1;
  for (int c = 0; c < 3; ++c) {
    g->min_b[c] = (int)floorf(mn[c] * inv);                        // :87-92
    const int max_b = (int)floorf(mx[c] * inv);
    g->div_b[c] = max_b - g->min_b[c] + 1;                         // :95
    vox *= (double)g->div_b[c];
  }
  g->wx = (g->div_b[0] + 31) >> 5;
  const double nw = (double)g->wx * g->div_b[1] * g->div_b[2];
  g->status = (nw > (double)kNdtMaxWords || !(vox > 0)) ? 7 : 0;
  g->nw = g->status ? 0 : (int)nw;
  g->nocc = 0;
}

__device__ __forceinline__ bool ndt_voxel_of(const NdtGridInfo* g, float x, float y, float z, int& i0, int& i1, int& i2) {
  // :218-220: static_cast<int>(floor(p * inverse_leaf_size) - static_cast<float>(min_b))
  i0 = (int)(floorf(x * g->inv) - (float)g->min_b[0]);
  i1 = (int)(floorf(y * g->inv) - (float)g->min_b[1]);
  i2 = (int)(floorf(z * g->inv) - (float)g->min_b[2]);
  return i0 >= 0 && i1 >= 0 && i2 >= 0 && i0 < g->div_b[0] && i1 < g->div_b[1] && i2 < g->div_b[2];
}

// one 1024-thread block: words = {bits, exclusive rank}; nocc
__global__ __launch_bounds__(1024) void ndt_voxel_rank(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.x];
  NdtGridInfo* g = d.info;
  const int nw = g->nw;
  __shared__ uint32_t s_w[17];
  uint32_t carry = 0;
  for (int t0 = 0; t0 < nw; t0 += 4096) {
    const int w = t0 + (int)threadIdx.x * 4;
    uint32_t v[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) if (w + k < nw) v[k] = d.bits[w + k];
    const uint32_t c = __popc(v[0]) + __popc(v[1]) + __popc(v[2]) + __popc(v[3]);
    uint32_t total;
    uint32_t run = carry + block_excl_scan(c, s_w, &total);
    for (int k = 0; k < 4; ++k) {
      if (w + k < nw) d.words[w + k] = make_uint2(v[k], run);
      run += __popc(v[k]);
    }
    carry += total;
  }
  if (threadIdx.x == 0) g->nocc = (int)carry;
}

// ---- sort-based build: (voxel code, point) pairs radix-sorted by the caller, then one pass marks the
// first point of every voxel (one atomicOr per VOXEL instead of one per point: 500 k points fall into a few
// thousand 1 m voxels) and one pass, after ndt_voxel_rank, records the voxel starts and gathers the points.
// (key = table << 33 | code: ONE radix sort orders the (voxel, point) pairs of every table of the batch, each table's run where
// its key_off says)
constexpr unsigned long long kNdtCodeMask = (1ull << 33) - 1ull;
__global__ __launch_bounds__(256) void ndt_voxel_keys64(const NdtDev* __restrict__ devs, unsigned long long* keys, int32_t* vals) {
  const NdtDev d = devs[blockIdx.y];
  const NdtGridInfo* g = d.info;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.nt) return;
  unsigned long long code = 0xffffffffull;                          // non-finite / out-of-box points sort last (:209-213)
  if (!g->status) {
    const float4 p = d.tgt[j];
    int i0, i1, i2;
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && ndt_voxel_of(g, p.x, p.y, p.z, i0, i1, i2))
      code = ((unsigned long long)((i2 * g->div_b[1] + i1) * g->wx + (i0 >> 5)) << 5) | (unsigned long long)(i0 & 31);
  }
  keys[d.key_off + j] = ((unsigned long long)blockIdx.y << 33) | code;
  vals[d.key_off + j] = j;
}

__global__ __launch_bounds__(256) void ndt_voxel_heads(const NdtDev* __restrict__ devs, const unsigned long long* keys) {
  const NdtDev d = devs[blockIdx.y];
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.nt) return;
  keys += d.key_off;
  const unsigned long long c = keys[s] & kNdtCodeMask;
  if (c == 0xffffffffull) return;
  if (s == 0 || (keys[s - 1] & kNdtCodeMask) != c) atomicOr(&d.bits[(uint32_t)(c >> 5)], 1u << (uint32_t)(c & 31));
}

__global__ __launch_bounds__(256) void ndt_voxel_starts(const NdtDev* __restrict__ devs, const unsigned long long* keys, const int32_t* vals) {
  const NdtDev d = devs[blockIdx.y];
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.nt) return;
  keys += d.key_off; vals += d.key_off;
  const unsigned long long c = keys[s] & kNdtCodeMask;
  const bool valid = c != 0xffffffffull;
  if (valid) {
    const int j = vals[s];
    float4 p = d.tgt[j];
    p.w = __int_as_float(j);
    d.vpts[s] = p;
    if (s == 0 || (keys[s - 1] & kNdtCodeMask) != c) {
      const uint2 wd = d.words[(uint32_t)(c >> 5)];
      d.vstart[wd.y + __popc(wd.x & ((1u << (uint32_t)(c & 31)) - 1u))] = (uint32_t)s;
    }
  }
  // one past the last valid point closes the last voxel
  if (valid && (s == d.nt - 1 || (keys[s + 1] & kNdtCodeMask) == 0xffffffffull)) d.vstart[d.info->nocc] = (uint32_t)s + 1u;
  if (s == 0 && !valid) d.vstart[0] = 0u;
}

__device__ void jacobi_eig3(double* A, double* V, double* w) {     // symmetric 3x3, cyclic Jacobi
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
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
        for (int k = 0; k < 3; ++k) { const double vkp = V[3 * k + p], vkq = V[3 * k + q]; V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[4 * i];
  // ascending order like SelfAdjointEigenSolver
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 2 - a; ++c)
      if (w[c] > w[c + 1]) {
        const double t = w[c]; w[c] = w[c + 1]; w[c + 1] = t;
        for (int k = 0; k < 3; ++k) { const double u = V[3 * k + c]; V[3 * k + c] = V[3 * k + c + 1]; V[3 * k + c + 1] = u; }
      }
}

// One wave per occupied voxel: sums over its points, then lane 0 finishes the Leaf (:282-367).
__device__ __forceinline__ void ndt_voxel_stats_one(const NdtDev& d, int v, int lane);
__global__ __launch_bounds__(256) void ndt_voxel_stats(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  // (the host does not know nocc: a grid sized by the point count is millions of empty workgroups for a batch of dense
  // submaps -- 4.9 ms of dispatch for 64 x 500 k points -- so a bounded grid strides over the occupied voxels instead)
  const int nocc = d.info->nocc;
  for (int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); v < nocc; v += gridDim.x * (blockDim.x >> 6)) ndt_voxel_stats_one(d, v, lane);
}
__device__ __forceinline__ void ndt_voxel_stats_one(const NdtDev& d, int v, int lane) {
  const uint32_t j0 = d.vstart[v], j1 = d.vstart[v + 1];
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float cs[3] = {0, 0, 0};
  auto add = [&](const float4 p) {
    const double x = p.x, y = p.y, z = p.z;
    s[0] += x; s[1] += y; s[2] += z;                                 // leaf.mean_ += pt3d, :233
    s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;   // leaf.cov_ += pt pt^T, :235
    cs[0] += p.x; cs[1] += p.y; cs[2] += p.z;                        // float centroid, :241
  };
  // four loads in flight per lane, added in the order of the plain loop (a 1 m voxel of a dense submap holds thousands of points,
  // and with one load per trip the wave that owns it waited a memory latency per 64 of them)
  uint32_t j = j0 + lane;
  for (; j + 192 < j1; j += 256) {
    const float4 p0 = d.vpts[j], p1 = d.vpts[j + 64], p2 = d.vpts[j + 128], p3 = d.vpts[j + 192];
    add(p0); add(p1); add(p2); add(p3);
  }
  for (; j < j1; j += 64) add(d.vpts[j]);
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    for (int off = 32; off > 0; off >>= 1) cs[k] += __shfl_down(cs[k], off, 64);
  if (lane != 0) return;
  NdtVoxel& o = d.vox[v];
  const int n = (int)(j1 - j0);
  const double nn = n;
  for (int k = 0; k < 3; ++k) o.centroid[k] = cs[k] / (float)n;      // :289
  const double mean[3] = {s[0] / nn, s[1] / nn, s[2] / nn};          // :293
  for (int k = 0; k < 3; ++k) o.mean[k] = mean[k];
  for (int k = 0; k < 6; ++k) { o.icov[k] = 0.f; d.icovd[(size_t)v * 6 + k] = 0.0; }
  o.n = n;
  if (n < d.min_points) return;                                      // :297 (not searchable)
  // cov_ started as identity (Leaf ctor), :329-330
  double C[9];
  const double acc[9] = {1.0 + s[3], s[4], s[5], s[4], 1.0 + s[6], s[7], s[5], s[7], 1.0 + s[8]};
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      C[3 * a + b] = ((acc[3 * a + b] - 2.0 * (s[a] * mean[b])) / nn + mean[a] * mean[b]) * ((nn - 1.0) / nn);
  double E[9], V[9], w[3];
  for (int k = 0; k < 9; ++k) E[k] = 0.5 * (C[k] + C[3 * (k % 3) + k / 3]);
  jacobi_eig3(E, V, w);
  if (w[0] < 0 || w[1] < 0 || w[2] <= 0) { o.n = -1; return; }       // :337-341
  const double m = d.eig_mult * w[2];                                // :345
  if (w[0] < m) {                                                    // :346-356
    w[0] = m;
    if (w[1] < m) w[1] = m;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        C[3 * a + b] = V[3 * a] * w[0] * V[3 * b] + V[3 * a + 1] * w[1] * V[3 * b + 1] + V[3 * a + 2] * w[2] * V[3 * b + 2];
  }
  // icov = cov^-1 (cofactors), :359
  const double c00 = C[4] * C[8] - C[5] * C[7], c01 = C[5] * C[6] - C[3] * C[8], c02 = C[3] * C[7] - C[4] * C[6];
  const double det = C[0] * c00 + C[1] * c01 + C[2] * c02;
  const double id = 1.0 / det;
  const double I[9] = {c00 * id, (C[2] * C[7] - C[1] * C[8]) * id, (C[1] * C[5] - C[2] * C[4]) * id,
                       c01 * id, (C[0] * C[8] - C[2] * C[6]) * id, (C[2] * C[3] - C[0] * C[5]) * id,
                       c02 * id, (C[1] * C[6] - C[0] * C[7]) * id, (C[0] * C[4] - C[1] * C[3]) * id};
  bool bad = false;
  for (int k = 0; k < 9; ++k) bad |= !isfinite(I[k]);
  if (bad) { o.n = -1; return; }                                     // :360-364
  o.icov[0] = (float)I[0]; o.icov[1] = (float)I[1]; o.icov[2] = (float)I[2];
  o.icov[3] = (float)I[4]; o.icov[4] = (float)I[5]; o.icov[5] = (float)I[8];
  double* oi = d.icovd + (size_t)v * 6;
  oi[0] = I[0]; oi[1] = I[1]; oi[2] = I[2]; oi[3] = I[4]; oi[4] = I[5]; oi[5] = I[8];
}

// ------------------------------------------------------------------------------------------
// derivatives
// ------------------------------------------------------------------------------------------
// R = float restates pclomp (ndt_omp_impl.hpp: Matrix<float,4,6> math); R = double restates stock
// pcl::NormalDistributionsTransform (PCL 1.8.1 ndt.hpp: the same formulas on Vector3d / Matrix3d), which
// registrators/ndt_gicp.cc:38-41,84-89 uses.
// ONE = the grid has a thread for every source point (always, up to 524 288 points): the per-point sums of the reference
// (score_pt, g_pt, h_pt: summed per point first, then added to the totals) are then the thread's totals themselves and
// need no registers of their own -- 43 doubles fewer per lane.
// grid = (workgroups per table, evaluations): evaluation e is pose poses[e] against table devs[active[e]] -- the K Aligns of a
// batch advance in lock-step on the host and every round's evaluations are ONE launch.
// A round of at most kNdtArgPoses evaluations (every round of a single Align) carries its poses and tables in the launch's own
// arguments: no copy to the device before the launch.  Larger rounds read them from device arrays.
constexpr int kNdtArgPoses = 4;
struct NdtPoseArgs {
  int32_t n;
  int32_t active[kNdtArgPoses];
  int32_t pad[3];
  NdtPose p[kNdtArgPoses];
};

template <typename R, bool ONE>
__device__ __forceinline__ void ndt_derivatives_body(const NdtDev& d, const NdtPose& P) {
  constexpr bool kDouble = sizeof(R) == 8;
  const NdtGridInfo* g = d.info;
  double acc[43];
#pragma unroll
  for (int k = 0; k < 43; ++k) acc[k] = 0.0;
  double pairs = 0;
  __shared__ uint32_t s_nb[27][kNdtDerivThreads];          // per-thread list of occupied neighbour voxels (slots)
  const R gd2 = kDouble ? (R)P.d2d : (R)P.d2;
  for (int i = blockIdx.x * kNdtDerivThreads + threadIdx.x; i < d.ns; i += gridDim.x * kNdtDerivThreads) {
    const float4 s = d.src[i];
    if (!(isfinite(s.x) && isfinite(s.y) && isfinite(s.z))) continue;
    // pcl::transformPointCloud with the float final_transformation_
    const float tx = P.T[0] * s.x + P.T[1] * s.y + P.T[2] * s.z + P.T[3];
    const float ty = P.T[4] * s.x + P.T[5] * s.y + P.T[6] * s.z + P.T[7];
    const float tz = P.T[8] * s.x + P.T[9] * s.y + P.T[10] * s.z + P.T[11];
    // voxel of the transformed point; KDTREE radius search == 27-stencil filtered by centroid distance
    const int c0 = (int)(floorf(tx * g->inv) - (float)g->min_b[0]);
    const int c1 = (int)(floorf(ty * g->inv) - (float)g->min_b[1]);
    const int c2 = (int)(floorf(tz * g->inv) - (float)g->min_b[2]);
    // point gradient (4x6 float): identity + 8 angular entries, :397-412
    R pg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      pg[r] = kDouble ? (R)(P.j_angd[r][0] * (double)s.x + P.j_angd[r][1] * (double)s.y + P.j_angd[r][2] * (double)s.z)
                      : (R)(P.j_ang[r][0] * s.x + P.j_ang[r][1] * s.y + P.j_ang[r][2] * s.z);
    R ph[15];
    if (P.compute_hessian) {
#pragma unroll
      for (int r = 0; r < 15; ++r)                                                                             // :416
        ph[r] = kDouble ? (R)(P.h_angd[r][0] * (double)s.x + P.h_angd[r][1] * (double)s.y + P.h_angd[r][2] * (double)s.z)
                        : (R)(P.h_ang[r][0] * s.x + P.h_ang[r][1] * s.y + P.h_ang[r][2] * s.z);
    }
    double pt_sums[ONE ? 1 : 43];
    double* const pt = ONE ? acc : pt_sums;                 // [0] score, [1..6] gradient, [7..42] hessian of this point
    if (!ONE) {
#pragma unroll
      for (int k = 0; k < 43; ++k) pt_sums[ONE ? 0 : k] = 0.0;
    }
    // The occupied voxels among the 27 neighbours, in the z, y, x order of the reference's loop.  The occupancy words of the
    // nine (z, y) rows are loaded together (a row's three cells lie in at most two words) and the slots parked in LDS; the
    // loop below then visits only the occupied neighbours (3.3 on average) instead of waiting for a word lookup per cell.
    int nnb = 0;
    {
      const int div0 = g->div_b[0], div1 = g->div_b[1], div2 = g->div_b[2];
      const int xa = min(max(c0 - 1, 0), div0 - 1), xb = min(max(c0 + 1, 0), div0 - 1);
      uint2 wa[9], wb[9];
      bool rowok[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int z = c2 + r / 3 - 1, y = c1 + r % 3 - 1;
        rowok[r] = z >= 0 && z < div2 && y >= 0 && y < div1;
        const int rowbase = rowok[r] ? (z * div1 + y) * g->wx : 0;
        wa[r] = d.words[rowbase + (xa >> 5)];
        wb[r] = d.words[rowbase + (xb >> 5)];
      }
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = c0 + dx;
          if (!rowok[r] || x < 0 || x >= div0) continue;
          const uint2 wd = (x >> 5) == (xa >> 5) ? wa[r] : wb[r];
          if (!((wd.x >> (x & 31)) & 1u)) continue;
          s_nb[nnb++][threadIdx.x] = wd.y + __popc(wd.x & ((1u << (x & 31)) - 1u));
        }
    }
    // (float arithmetic: the next neighbour's record is fetched while this one is processed; the double variant has no
    // registers to spare for that)
    uint32_t slot_next = nnb > 0 ? s_nb[0][threadIdx.x] : 0u;
    NdtVoxel vx_next;
    if (!kDouble && nnb > 0) vx_next = d.vox[slot_next];
    for (int kn = 0; kn < nnb; ++kn) {
      {
        {
          const uint32_t slot = slot_next;
          NdtVoxel vx;
          if (kDouble) vx = d.vox[slot]; else vx = vx_next;
          if (kn + 1 < nnb) {
            slot_next = s_nb[kn + 1][threadIdx.x];
            if (!kDouble) vx_next = d.vox[slot_next];
          }
          if (vx.n >= 0 && vx.n < d.min_points) continue;            // not in the centroid kd-tree
          const float ex = tx - vx.centroid[0], ey = ty - vx.centroid[1], ez = tz - vx.centroid[2];
          if (ex * ex + ey * ey + ez * ez > P.res2) continue;        // radiusSearch(x_trans, resolution_), :235
          pairs += 1.0;
          // x_trans - mean in double, then float (:253, :490)
          const R u0 = (R)((double)tx - vx.mean[0]), u1 = (R)((double)ty - vx.mean[1]), u2 = (R)((double)tz - vx.mean[2]);
          R cxx, cxy, cxz, cyy, cyz, czz;
          if (kDouble) {
            const double* ic = d.icovd + (size_t)slot * 6;
            cxx = (R)ic[0]; cxy = (R)ic[1]; cxz = (R)ic[2]; cyy = (R)ic[3]; cyz = (R)ic[4]; czz = (R)ic[5];
          } else {
            cxx = vx.icov[0]; cxy = vx.icov[1]; cxz = vx.icov[2]; cyy = vx.icov[3]; cyz = vx.icov[4]; czz = vx.icov[5];
          }
          // x_trans4 * c_inv4
          const R v0 = u0 * cxx + u1 * cxy + u2 * cxz;
          const R v1 = u0 * cxy + u1 * cyy + u2 * cyz;
          const R v2 = u0 * cxz + u1 * cyz + u2 * czz;
          const R q = u0 * v0 + u1 * v1 + u2 * v2;
          R e = kDouble ? (R)exp(-(double)gd2 * (double)q / 2) : (R)expf(-(float)gd2 * (float)q * 0.5f);   // :497
          const R score_inc = (R)(-P.d1d * (double)e);               // :499
          e = gd2 * e;                                               // :501
          if (e > (R)1 || e < (R)0 || e != e) continue;              // :504-505
          e = (R)(P.d1d * (double)e);                                // :508
          pt[0] += (double)score_inc;
          // columns of c_inv4 * point_gradient4: col 0..2 = columns of C; col 3..5 from the angular entries
          // J col3 = (0, pg0, pg1), col4 = (pg2, pg3, pg4), col5 = (pg5, pg6, pg7)
          R CJ[6][3];
          CJ[0][0] = cxx; CJ[0][1] = cxy; CJ[0][2] = cxz;
          CJ[1][0] = cxy; CJ[1][1] = cyy; CJ[1][2] = cyz;
          CJ[2][0] = cxz; CJ[2][1] = cyz; CJ[2][2] = czz;
          CJ[3][0] = cxy * pg[0] + cxz * pg[1]; CJ[3][1] = cyy * pg[0] + cyz * pg[1]; CJ[3][2] = cyz * pg[0] + czz * pg[1];
          CJ[4][0] = cxx * pg[2] + cxy * pg[3] + cxz * pg[4]; CJ[4][1] = cxy * pg[2] + cyy * pg[3] + cyz * pg[4]; CJ[4][2] = cxz * pg[2] + cyz * pg[3] + czz * pg[4];
          CJ[5][0] = cxx * pg[5] + cxy * pg[6] + cxz * pg[7]; CJ[5][1] = cxy * pg[5] + cyy * pg[6] + cyz * pg[7]; CJ[5][2] = cxz * pg[5] + cyz * pg[6] + czz * pg[7];
          R xCJ[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) xCJ[c] = u0 * CJ[c][0] + u1 * CJ[c][1] + u2 * CJ[c][2];     // :511
#pragma unroll
          for (int c = 0; c < 6; ++c) pt[1 + c] += (double)(e * xCJ[c]);                              // :513
          if (P.compute_hessian) {
            // J columns as 3-vectors
            R Jc[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, pg[0], pg[1]}, {pg[2], pg[3], pg[4]}, {pg[5], pg[6], pg[7]}};
            // x_trans4_x_c_inv4 . point_hessian block (i, j): only i, j in 3..5 are non-zero, :418-437
            // a=(0,ph0,ph1) b=(0,ph2,ph3) c=(0,ph4,ph5) d=(ph6,ph7,ph8) e=(ph9,ph10,ph11) f=(ph12,ph13,ph14)
            const R ha = v1 * ph[0] + v2 * ph[1], hb = v1 * ph[2] + v2 * ph[3], hc = v1 * ph[4] + v2 * ph[5];
            const R hd = v0 * ph[6] + v1 * ph[7] + v2 * ph[8], he = v0 * ph[9] + v1 * ph[10] + v2 * ph[11];
            const R hf = v0 * ph[12] + v1 * ph[13] + v2 * ph[14];
            const R xH[3][3] = {{ha, hb, hc}, {hb, hd, he}, {hc, he, hf}};     // [i-3][j-3]
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int c = 0; c < 6; ++c) {
                // point_gradient4.col(j) . (c_inv4 * point_gradient4.col(i))  -> (j, i) entry, :517, :529
                const R jcj = Jc[c][0] * CJ[a][0] + Jc[c][1] * CJ[a][1] + Jc[c][2] * CJ[a][2];
                const R hh = (a >= 3 && c >= 3) ? xH[a - 3][c - 3] : (R)0;
                pt[7 + 6 * a + c] += (double)(e * (-gd2 * xCJ[a] * xCJ[c] + hh + jcj));               // :527-529
              }
          }
        }
      }
    }
    if (!ONE) {
      acc[0] += pt[0];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[1 + c] += pt[1 + c];
      if (P.compute_hessian) {
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[7 + k] += pt[7 + k];
      }
    }
  }
  // block reduction -> partials
  __shared__ double s_red[kNdtDerivThreads / 64][kNdtDerivCols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 43; ++k) acc[k] = wave_sum_to_last(acc[k]);      // DPP row operations, total in lane 63
  pairs = wave_sum_to_last(pairs);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 43; ++k) s_red[wave][k] = acc[k];
    s_red[wave][43] = pairs;
  }
  __syncthreads();
  if (threadIdx.x < kNdtDerivCols) {
    double t = 0;
    for (int w = 0; w < kNdtDerivThreads / 64; ++w) t += s_red[w][threadIdx.x];
    d.partials[(size_t)blockIdx.x * kNdtDerivCols + threadIdx.x] = t;
  }
}
template <typename R, bool ONE>
__global__ __launch_bounds__(kNdtDerivThreads, ONE ? 2 : 1) void ndt_derivatives(const NdtDev* __restrict__ devs, const NdtPose* __restrict__ poses,
                                                                                    const int32_t* __restrict__ active) {
  const NdtDev d = devs[active[blockIdx.y]];
  ndt_derivatives_body<R, ONE>(d, poses[blockIdx.y]);
}
template <typename R, bool ONE>
__global__ __launch_bounds__(kNdtDerivThreads, ONE ? 2 : 1) void ndt_derivatives_args(const NdtDev* __restrict__ devs, const NdtPoseArgs A) {
  const NdtDev d = devs[A.active[blockIdx.y]];
  ndt_derivatives_body<R, ONE>(d, A.p[blockIdx.y]);
}

// 16 thread groups take every 16th block, then one thread per column folds the group sums:
// a fixed order, so repeated evaluations at the same pose are bitwise identical.  The sums also go straight into page-locked
// host memory (row e of out_host = evaluation e of the round): the host reads them after the stream's synchronise, no copy.
struct NdtActiveArgs { int32_t slot[kNdtArgPoses]; };
__device__ __forceinline__ void ndt_reduce_body(const NdtDev& d, int nblocks, double* __restrict__ out_host) {
  __shared__ double s_g[16][kNdtDerivCols];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (c < kNdtDerivCols) {
    double t = 0;
    for (int k = grp; k < nblocks; k += 16) t += d.partials[(size_t)k * kNdtDerivCols + c];
    s_g[grp][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < kNdtDerivCols) {
    double t = 0;
    for (int g = 0; g < 16; ++g) t += s_g[g][threadIdx.x];
    d.out[threadIdx.x] = t;
    out_host[(size_t)blockIdx.x * kNdtDerivCols + threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(16 * 64) void ndt_reduce(const NdtDev* __restrict__ devs, const int32_t* __restrict__ active, int nblocks, double* __restrict__ out_host) {
  const NdtDev d = devs[active[blockIdx.x]];
  ndt_reduce_body(d, nblocks, out_host);
}
__global__ __launch_bounds__(16 * 64) void ndt_reduce_args(const NdtDev* __restrict__ devs, const NdtActiveArgs A, int nblocks, double* __restrict__ out_host) {
  const NdtDev d = devs[A.slot[blockIdx.x]];
  ndt_reduce_body(d, nblocks, out_host);
}

// mean of the squared NN distances of slot 0 (pcl::Registration::getFitnessScore, ndt.cc:60)
// grid = (64, pairs): pair slot pair_base + blockIdx.y of the matcher's d2 array ([slots][ns_cap]), ns[blockIdx.y] valid entries
// (the sizes ride in the launch's arguments and the partial sums go straight into page-locked host memory: no copy either way)
constexpr int kFitnessArgPairs = 64;
struct FitnessArgs { int32_t ns[kFitnessArgPairs]; };
__global__ __launch_bounds__(256) void fitness_partial(const float* d2_all, size_t ns_cap, int pair_base, const FitnessArgs A, double* partials_all) {
  const float* d2 = d2_all + (size_t)(pair_base + blockIdx.y) * ns_cap;
  const int n = A.ns[blockIdx.y];
  double* partials = partials_all + (size_t)blockIdx.y * 128;
  double s = 0, c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = d2[i];
    if (__float_as_uint(v) < 0x7f800000u) { s += (double)v; c += 1.0; }
  }
  __shared__ double s_s[4], s_c[4];
  s = wave_sum(s); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { s_s[threadIdx.x >> 6] = s; s_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s_s[0] + s_s[1] + s_s[2] + s_s[3];
    partials[2 * blockIdx.x + 1] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
  }
}

}  // namespace smhip
