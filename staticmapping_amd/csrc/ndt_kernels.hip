// ndt_kernels.hip -- hand-written gfx950 kernels for registrators::Ndt (pclomp NDT).
//
// Reference lines are in /root/reference/registrators:
//   ndt_voxel_*      VoxelGridCovariance::applyFilter   pclomp/voxel_grid_covariance_omp_impl.hpp:49-370
//   ndt_derivatives  computeDerivatives + computePointDerivatives + updateDerivatives
//                                                       pclomp/ndt_omp_impl.hpp:180-284, 397-438, 483-535
//   ndt_ctl_step     the fold of the partial sums (:272-281) + the Newton / More-Thuente driver
//                                                       pclomp/ndt_omp_impl.hpp:81-171, 757-916
// As in the reference the per-neighbour math is float and the accumulation double.
#pragma once
#include "smhip_device.h"

namespace smhip {

constexpr int kNdtMaxWords = 1 << 20;        // 32 voxels per word: up to 32 Mi voxels in the dense box
constexpr int kNdtDerivThreads = 256;

struct NdtVoxel {          // 64 B record of one searchable voxel (n >= min_points)
  double mean[3];          // Leaf::mean_  (double)
  float icov[6];           // Leaf::icov_ cast to float: xx xy xz yy yz zz
  float centroid[3];       // Leaf::centroid (float)
  int32_t n;               // nr_points; -1 = eigen check failed (stays searchable with icov = 0)
};

struct NdtGridInfo {       // written by ndt_voxel_setup and the build kernels after it
  int32_t min_b[3];
  int32_t div_b[3];
  int32_t wx;              // words per x row
  int32_t nw;              // total words
  int32_t nocc;            // occupied voxels
  int32_t status;          // 0 ok, 7 = voxel box larger than the bit grid
  float inv;               // inverse leaf size (float, as PCL)
  float res;
  int32_t nvalid;          // target points inside the box (finite)
  int32_t nsub;            // (unused)
  int32_t nbig;            // voxels with more than kNdtBigVoxel points
  uint32_t nlist;          // fitness search: queries the first pass left to the second / the second to the sweep
  uint32_t nleft;
  float marg;              // what the float rounding of p * inv can move a point across a lattice plane, in cells
  int32_t pad_[2];
};

// The fine levels of the table.  Every voxel is cut into 4 x 4 x 4 mid cells and each of those into 4 x 4 x 4 fine cells (1/16 of a
// voxel: 6.25 cm at the reference's 1 m; coordinates from the fractional part of p * inv), and a voxel's points lie sorted by
// mid cell, then fine cell ((z * 4 + y) * 4 + x), then caller index -- so a voxel, a mid cell, a fine cell and an x-row of fine cells are
// each ONE run of vpts.  The NDT statistics do not care (any fixed order of a voxel's points serves); the exact 1-NN search of the
// fitness score finds a mid cell through an open-addressing hash table (key: voxel code | mid cell coordinates) whose entry
// carries the cell's run, the occupancy mask of its 64 fine cells and, through fpos[start + rank], where each occupied fine cell
// begins.  Neighbouring queries probe the same few entries: the table is read through the caches, not at random.
struct alignas(32) NdtCell {
  unsigned long long key;  // ~0 = empty
  uint32_t start, end;     // the mid cell's points: vpts[start, end)
  unsigned long long fmask;
  unsigned long long pad_;
};
__device__ __forceinline__ uint32_t ndt_cell_hash(unsigned long long key, int log2cap) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap)); }
__device__ __forceinline__ unsigned long long ndt_mid_key(uint32_t code, int mz, int my, int mx) {
  return ((unsigned long long)code << 6) | (unsigned long long)((mz << 4) | (my << 2) | mx);
}

constexpr int kNdtBoxBlocks = 128;           // workgroups per target of the bounding-box pass
constexpr int kNdtBigVoxel = 2048;           // voxels with more points get a whole 1024-thread workgroup for their sums

// One voxel table (one pair slot's target).  The kernels take a device ARRAY of these and pick theirs by block index: a batch
// of K Aligns builds its K tables and evaluates its K derivative sets in single launches (grid.y = pair).
struct NdtDev {
  int32_t nt, ns;
  int32_t min_points;
  float eig_mult;
  const float4* tgt;       // raw target points
  const float4* src;       // raw source points (Morton order)
  NdtGridInfo* info;
  float* bbox;             // [kNdtBoxBlocks][8] min xyz, max xyz of a stripe of the target
  uint32_t* bits;          // [kNdtMaxWords]
  uint2* words;            // [kNdtMaxWords]
  uint32_t* vstart;        // [nt + 1]
  float4* vpts;            // [nt] points sorted by voxel, mid cell, fine cell, caller index (w)
  NdtVoxel* vox;           // [nt] one record per occupied voxel
  NdtCell* cells;          // [1 << log2cells] hash table of the occupied mid cells
  uint32_t* fpos;          // [nt] fpos[start of a mid cell + r] = where its r-th occupied fine cell begins
  uint32_t* big;           // [nt / kNdtBigVoxel + 1] the crowded voxels
  double* vsum;            // [nt][12] a voxel's raw sums (x y z, xx xy xz yy yz zz, float centroid sums) between the summing and the finishing pass
  float4* vbox;            // [nt] lower corner of every occupied voxel in scaled coordinates (p * inv): the far search prunes by boxes
  uint32_t* qlist;         // [ns] fitness search: queries the first pass hands to the second
  uint32_t* qleft;         // [ns] ... and the second to the sweep
  double* icovd;           // [nt][6] Leaf::icov_ in double: xx xy xz yy yz zz (stock PCL path reads these)
  double* partials;        // [ceil(ns_cap / 256)][kNdtCols]
  double* out;             // [kNdtOutCols]
  float* fit_d2;           // [ns] squared distance to the nearest target point (fitness search)
  int32_t key_off;         // where this table's (key, point) pairs start in the batch's sort arrays
  int32_t key_bits;        // bits of slot << 12 | mid cell << 6 | fine cell in a sort key (the table number sits above them)
  int32_t log2cells;       // size of `cells`: at least two entries per target point (never more than half full)
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
// bbox -> setup -> mark (occupancy bit per voxel) -> rank (popcount dictionary: bit -> slot) -> keys (slot | mid cell | fine cell)
// -> ONE radix sort of every table's (key, point) pairs -> heads (points gathered into sorted order, voxel starts, the cells' runs
// opened in the hash table) -> tails (the runs closed, list of crowded voxels) -> stats / stats_big (the Leaf of every voxel).
__global__ __launch_bounds__(256) void ndt_bbox(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < d.nt; j += gridDim.x * blockDim.x) {
    const float4 p = d.tgt[j];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s_mn[4][3], s_mx[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
  if (lane == 0) for (int c = 0; c < 3; ++c) { s_mn[wave][c] = mn[c]; s_mx[wave][c] = mx[c]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    d.bbox[8 * blockIdx.x + c] = fminf(fminf(s_mn[0][c], s_mn[1][c]), fminf(s_mn[2][c], s_mn[3][c]));
    d.bbox[8 * blockIdx.x + 4 + c] = fmaxf(fmaxf(s_mx[0][c], s_mx[1][c]), fmaxf(s_mx[2][c], s_mx[3][c]));
  }
}

__global__ void ndt_voxel_setup(const NdtDev* __restrict__ devs, float leaf) {
  const NdtDev d = devs[blockIdx.x];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = threadIdx.x; k < kNdtBoxBlocks; k += 64)
    for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], d.bbox[8 * k + c]); mx[c] = fmaxf(mx[c], d.bbox[8 * k + 4 + c]); }
  for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
  if (threadIdx.x != 0) return;
  NdtGridInfo* g = d.info;
  const float inv = 1.0f / leaf;                                   // inverse_leaf_size_
  g->inv = inv; g->res = leaf;
  double vox = 1;
  float amax = 0.f;
  for (int c = 0; c < 3; ++c) {
    g->min_b[c] = (int)floorf(mn[c] * inv);                        // :87-92
    const int max_b = (int)floorf(mx[c] * inv);
    g->div_b[c] = max_b - g->min_b[c] + 1;                         // :95
    vox *= (double)g->div_b[c];
    amax = fmaxf(amax, fmaxf(fabsf(mn[c] * inv), fabsf(mx[c] * inv)));
  }
  g->wx = (g->div_b[0] + 31) >> 5;
  const double nw = (double)g->wx * g->div_b[1] * g->div_b[2];
  g->status = (nw > (double)kNdtMaxWords || !(vox > 0)) ? 7 : 0;
  g->nw = g->status ? 0 : (int)nw;
  g->nocc = 0; g->nvalid = 0; g->nsub = 0; g->nbig = 0; g->nlist = 0; g->nleft = 0;
  // p * inv is one float rounding: a lattice plane can be off by 2^-24 |p * inv| for a target point and as much for a query
  g->marg = 4.0f * 5.9604645e-08f * (amax + 2.0f);
}

__device__ __forceinline__ bool ndt_voxel_of(const NdtGridInfo* g, float x, float y, float z, int& i0, int& i1, int& i2) {
  // :218-220: static_cast<int>(floor(p * inverse_leaf_size) - static_cast<float>(min_b))
  i0 = (int)(floorf(x * g->inv) - (float)g->min_b[0]);
  i1 = (int)(floorf(y * g->inv) - (float)g->min_b[1]);
  i2 = (int)(floorf(z * g->inv) - (float)g->min_b[2]);
  return i0 >= 0 && i1 >= 0 && i2 >= 0 && i0 < g->div_b[0] && i1 < g->div_b[1] && i2 < g->div_b[2];
}
// the fine coordinate of a point inside its voxel: floor(16 frac(p * inv)) per axis -- exact float operations on the one rounded
// product; its upper two bits are the mid cell's coordinate, the lower two the fine cell's inside the mid cell
__device__ __forceinline__ int ndt_fine_axis(float p, float inv) {
  const float ps = p * inv;
  return min(15, (int)((ps - floorf(ps)) * 16.0f));
}
__device__ __forceinline__ uint32_t ndt_voxel_code(const NdtGridInfo* g, const float4 p) {     // word << 5 | bit, or ~0
  int i0, i1, i2;
  if (g->status || !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) || !ndt_voxel_of(g, p.x, p.y, p.z, i0, i1, i2)) return 0xffffffffu;   // :209-213
  return ((uint32_t)((i2 * g->div_b[1] + i1) * g->wx + (i0 >> 5)) << 5) | (uint32_t)(i0 & 31);
}

// 500 k points of a submap fill a few thousand voxels -- a dozen cache lines of the bit grid -- and one atomicOr per point queues
// up on them (a look at the word first is no better: device-scope loads of the same few lines queue up just the same; and the points
// arrive interleaved over the 64 rings, so neighbours in the array are no neighbours in space and nothing combines inside a wave).
// A workgroup of 1024 points holds ~100 distinct voxels: each point enters its code into an LDS set, and only the one that finds
// its place empty sends the atomicOr -- a tenth of the traffic on the hot lines.
constexpr int kNdtMarkThreads = 1024;
__global__ __launch_bounds__(kNdtMarkThreads) void ndt_voxel_mark(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  constexpr int kSet = 1024;
  __shared__ uint32_t s_set[kSet];
  s_set[threadIdx.x] = 0xffffffffu;
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.nt) return;
  const uint32_t code = ndt_voxel_code(d.info, d.tgt[j]);
  if (code == 0xffffffffu) return;
  uint32_t hpos = (code * 0x9E3779B1u) >> 22;
  for (int t = 0; t < 16; ++t) {
    const uint32_t was = atomicCAS(&s_set[hpos], 0xffffffffu, code);
    if (was == code) return;                               // another point of the workgroup speaks for this voxel
    if (was == 0xffffffffu) break;
    hpos = (hpos + 1u) & (kSet - 1);
  }
  atomicOr(&d.bits[code >> 5], 1u << (code & 31));         // first of its voxel here (or no room in its probe window)
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

// key = table << key_bits | slot << 12 | mid cell << 6 | fine cell; points outside the box (or not finite) get the table's largest key and sort last
__global__ __launch_bounds__(256) void ndt_voxel_keys64(const NdtDev* __restrict__ devs, unsigned long long* keys, int32_t* vals) {
  const NdtDev d = devs[blockIdx.y];
  const NdtGridInfo* g = d.info;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.nt) return;
  const float4 p = d.tgt[j];
  const uint32_t code = ndt_voxel_code(g, p);
  unsigned long long low = (1ull << d.key_bits) - 1ull;
  if (code != 0xffffffffu) {
    const uint2 wd = d.words[code >> 5];
    const uint32_t slot = wd.y + __popc(wd.x & ((1u << (code & 31)) - 1u));
    const int fx = ndt_fine_axis(p.x, g->inv), fy = ndt_fine_axis(p.y, g->inv), fz = ndt_fine_axis(p.z, g->inv);
    const uint32_t mid = (uint32_t)((((fz >> 2) * 4 + (fy >> 2)) * 4 + (fx >> 2)));
    const uint32_t fin = (uint32_t)((((fz & 3) * 4 + (fy & 3)) * 4 + (fx & 3)));
    low = ((unsigned long long)slot << 12) | (mid << 6) | fin;
  }
  keys[d.key_off + j] = ((unsigned long long)blockIdx.y << d.key_bits) | low;
  vals[d.key_off + j] = j;
}

// open-addressing insert / find of a cell's entry (linear probing; the table is never more than 2/3 full)
__device__ __forceinline__ NdtCell* ndt_cell_insert(const NdtDev& d, unsigned long long key) {
  const uint32_t maskc = (1u << d.log2cells) - 1u;
  uint32_t hpos = ndt_cell_hash(key, d.log2cells);
  for (;;) {
    const unsigned long long was = atomicCAS(&d.cells[hpos].key, ~0ull, key);
    if (was == ~0ull || was == key) return d.cells + hpos;
    hpos = (hpos + 1u) & maskc;
  }
}
__device__ __forceinline__ const NdtCell* ndt_cell_find(const NdtDev& d, unsigned long long key) {
  const uint32_t maskc = (1u << d.log2cells) - 1u;
  uint32_t hpos = ndt_cell_hash(key, d.log2cells);
  for (;;) {
    const unsigned long long k = d.cells[hpos].key;
    if (k == key) return d.cells + hpos;
    if (k == ~0ull) return nullptr;
    hpos = (hpos + 1u) & maskc;
  }
}
// the table starts empty: the mid cells' hash table over the part this target's size makes it use, and the words of the bit grid
// its box covers (known since ndt_voxel_setup)
__global__ __launch_bounds__(256) void ndt_cells_clear(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  const size_t n = (size_t)1 << d.log2cells;
  const int nw = d.info->nw;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nw; k += gridDim.x * blockDim.x) d.bits[k] = 0u;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    NdtCell e;
    e.key = ~0ull; e.start = 0u; e.end = 0u; e.fmask = 0ull; e.pad_ = 0ull;
    d.cells[k] = e;
  }
}

// the key of a sorted point's mid cell and the number of its fine cell inside it, from the point itself
__device__ __forceinline__ unsigned long long ndt_point_cell(const NdtGridInfo* g, const float4 p, int& fine) {
  const uint32_t code = ndt_voxel_code(g, p);
  const int fx = ndt_fine_axis(p.x, g->inv), fy = ndt_fine_axis(p.y, g->inv), fz = ndt_fine_axis(p.z, g->inv);
  fine = ((fz & 3) * 4 + (fy & 3)) * 4 + (fx & 3);
  return ndt_mid_key(code, fz >> 2, fy >> 2, fx >> 2);
}

// sorted order: every thread gathers its point; the first point of a mid cell walks the cell's run of keys (ten points on average),
// collects the occupancy mask of its fine cells and where each begins, and enters the cell into the hash table; the first point of a
// voxel records the voxel's start and box.  (One table insert per mid cell: with one per fine cell -- every second point of a
// submap -- the inserts and their mask updates were the build's second largest cost.)
__global__ __launch_bounds__(256) void ndt_voxel_heads(const NdtDev* __restrict__ devs, const unsigned long long* keys, const int32_t* vals) {
  const NdtDev d = devs[blockIdx.y];
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.nt) return;
  keys += d.key_off; vals += d.key_off;
  const unsigned long long lowmask = (1ull << d.key_bits) - 1ull;
  const unsigned long long k = keys[s] & lowmask;
  const bool valid = k != lowmask;
  if (valid) {
    const int j = vals[s];
    float4 p = d.tgt[j];
    p.w = __int_as_float(j);
    d.vpts[s] = p;
    const unsigned long long kp = s ? (keys[s - 1] & lowmask) : ~0ull;
    if (s == 0 || (kp >> 6) != (k >> 6)) {
      // the mid cell's run: keys[s ...] while the upper bits stay
      unsigned long long fm = 0ull, prev = ~0ull;
      uint32_t nf = 0, t = (uint32_t)s;
      bool more = true;
      while (more) {
        unsigned long long kk[4];                          // four keys in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) kk[u] = (t + u < (uint32_t)d.nt) ? (keys[t + u] & lowmask) : lowmask;
        const uint32_t t0 = t;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (more && (kk[u] >> 6) == (k >> 6)) {
            if (kk[u] != prev) { fm |= 1ull << (uint32_t)(kk[u] & 63); d.fpos[(uint32_t)s + nf] = t0 + u; ++nf; prev = kk[u]; }
            t = t0 + u + 1;
          } else more = false;
        }
      }
      int fine;
      NdtCell* e = ndt_cell_insert(d, ndt_point_cell(d.info, p, fine));
      e->start = (uint32_t)s; e->end = t; e->fmask = fm;
      if (s == 0 || (kp >> 12) != (k >> 12)) {
        d.vstart[(uint32_t)(k >> 12)] = (uint32_t)s;
        d.vbox[(uint32_t)(k >> 12)] = make_float4(floorf(p.x * d.info->inv), floorf(p.y * d.info->inv), floorf(p.z * d.info->inv), 0.f);
      }
    }
  }
  // one past the last valid point closes the last voxel
  if (valid && (s == d.nt - 1 || (keys[s + 1] & lowmask) == lowmask)) { d.vstart[d.info->nocc] = (uint32_t)s + 1u; d.info->nvalid = s + 1; }
  if (s == 0 && !valid) { d.vstart[0] = 0u; d.info->nvalid = 0; }
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

// One wave per occupied voxel: sums over its points, then lane 0 finishes the Leaf (:282-367).  A crowded voxel (a 1 m cell
// of a dense submap next to the sensor holds 10 000+ points, and the launch lasted as long as the one wave that owned it) is
// left to ndt_voxel_stats_big: a whole 1024-thread workgroup.  Which kernel sums a voxel follows from its size alone.
struct NdtVoxelSums { double s[9]; float cs[3]; };
__device__ __forceinline__ void ndt_voxel_add(NdtVoxelSums& a, const float4 p) {
  const double x = p.x, y = p.y, z = p.z;
  a.s[0] += x; a.s[1] += y; a.s[2] += z;                                 // leaf.mean_ += pt3d, :233
  a.s[3] += x * x; a.s[4] += x * y; a.s[5] += x * z; a.s[6] += y * y; a.s[7] += y * z; a.s[8] += z * z;   // leaf.cov_ += pt pt^T, :235
  a.cs[0] += p.x; a.cs[1] += p.y; a.cs[2] += p.z;                        // float centroid, :241
}
// a thread's share of the points [j0, j1): every `stride`-th from j0 + first, four loads in flight, added in the plain loop's order
__device__ __forceinline__ void ndt_voxel_walk(NdtVoxelSums& a, const float4* __restrict__ vpts, uint32_t j0, uint32_t j1, uint32_t first, uint32_t stride) {
  uint32_t j = j0 + first;
  for (; j + 3 * stride < j1; j += 4 * stride) {
    const float4 p0 = vpts[j], p1 = vpts[j + stride], p2 = vpts[j + 2 * stride], p3 = vpts[j + 3 * stride];
    ndt_voxel_add(a, p0); ndt_voxel_add(a, p1); ndt_voxel_add(a, p2); ndt_voxel_add(a, p3);
  }
  for (; j < j1; j += stride) ndt_voxel_add(a, vpts[j]);
}
__device__ __forceinline__ void ndt_voxel_store_sums(const NdtDev& d, int v, const double* s, const float* cs) {
  double* o = d.vsum + (size_t)v * 12;
  for (int k = 0; k < 9; ++k) o[k] = s[k];
  for (int k = 0; k < 3; ++k) o[9 + k] = (double)cs[k];                // (a float's double is exact)
}
__global__ __launch_bounds__(256) void ndt_voxel_stats(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  // (the host does not know nocc: a grid sized by the point count is millions of empty workgroups for a batch of dense
  // submaps -- 4.9 ms of dispatch for 64 x 500 k points -- so a bounded grid strides over the occupied voxels instead)
  const int nocc = d.info->nocc;
  for (int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); v < nocc; v += gridDim.x * (blockDim.x >> 6)) {
    const uint32_t j0 = d.vstart[v], j1 = d.vstart[v + 1];
    if (j1 - j0 > (uint32_t)kNdtBigVoxel) {
      if (lane == 0) d.big[atomicAdd((uint32_t*)&d.info->nbig, 1u)] = (uint32_t)v;
      continue;
    }
    NdtVoxelSums a{};
    ndt_voxel_walk(a, d.vpts, j0, j1, lane, 64);
#pragma unroll
    for (int k = 0; k < 9; ++k) a.s[k] = wave_sum(a.s[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      for (int off = 32; off > 0; off >>= 1) a.cs[k] += __shfl_down(a.cs[k], off, 64);
    if (lane == 0) ndt_voxel_store_sums(d, v, a.s, a.cs);
  }
}
__global__ __launch_bounds__(1024) void ndt_voxel_stats_big(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double s_s[16][9];
  __shared__ float s_c[16][3];
  const int nbig = d.info->nbig;
  for (int k = blockIdx.x; k < nbig; k += gridDim.x) {
    const int v = (int)d.big[k];
    const uint32_t j0 = d.vstart[v], j1 = d.vstart[v + 1];
    NdtVoxelSums a{};
    ndt_voxel_walk(a, d.vpts, j0, j1, threadIdx.x, 1024);
#pragma unroll
    for (int c = 0; c < 9; ++c) a.s[c] = wave_sum(a.s[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      for (int off = 32; off > 0; off >>= 1) a.cs[c] += __shfl_down(a.cs[c], off, 64);
    if (lane == 0) {
      for (int c = 0; c < 9; ++c) s_s[wave][c] = a.s[c];
      for (int c = 0; c < 3; ++c) s_c[wave][c] = a.cs[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t[9]; float tc[3];
      for (int c = 0; c < 9; ++c) { double r = 0; for (int w = 0; w < 16; ++w) r += s_s[w][c]; t[c] = r; }
      for (int c = 0; c < 3; ++c) { float r = 0; for (int w = 0; w < 16; ++w) r += s_c[w][c]; tc[c] = r; }
      ndt_voxel_store_sums(d, v, t, tc);
    }
    __syncthreads();
  }
}
// a thread per voxel finishes its Leaf from the sums (:282-367): with a wave per voxel this serial tail -- a Jacobi
// eigen-decomposition in double on one lane -- was most of the summing kernel's time
__device__ __forceinline__ void ndt_voxel_finish(const NdtDev& d, int v, const double* s, const float* cs);
__global__ __launch_bounds__(64) void ndt_voxel_leaves(const NdtDev* __restrict__ devs) {
  const NdtDev d = devs[blockIdx.y];
  const int nocc = d.info->nocc;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nocc; v += gridDim.x * blockDim.x) {
    const double* in = d.vsum + (size_t)v * 12;
    double s[9]; float cs[3];
    for (int k = 0; k < 9; ++k) s[k] = in[k];
    for (int k = 0; k < 3; ++k) cs[k] = (float)in[9 + k];
    ndt_voxel_finish(d, v, s, cs);
  }
}
__device__ __forceinline__ void ndt_voxel_finish(const NdtDev& d, int v, const double* s, const float* cs) {
  const uint32_t j0 = d.vstart[v], j1 = d.vstart[v + 1];
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
//
// Shape of the kernel.  A wave owns 64 source points (four granules of 16 Morton-consecutive points taken a quarter of the cloud
// apart, so every wave gets the cloud's average neighbour count); its four phases meet no other wave until the final fold.
//   A  (a lane per POINT): transform, the occupancy words of the nine (z, y) rows around its voxel (a row's three cells lie in
//      at most two words: 18 loads in flight), one (point, voxel slot) entry per OCCUPIED neighbour voxel appended to the wave's
//      list in LDS (order: point, then the reference's z, y, x neighbour order).
//   B1 (a lane per ENTRY): the entries the reference's radiusSearch would return -- voxel in the centroid kd-tree, centroid within
//      resolution_ of the point (:235; KDTREE mode == the 27-stencil filtered by centroid distance) -- kept, compacted in place;
//      a dense submap has 8-10 occupied voxels around a point of which 3 pass.  16-byte gathers (centroid + n), four in flight.
//   B2 (a lane per POINT again, over its kept voxels): the 48-byte rest of the voxel record (the hash-probed gather), then only what
//      depends on the voxel: x' = x_trans - mean, v = C x', e = d1 d2 exp(-d2/2 x'.v) with the reference's validity rule
//      (:497-508), and the four sums  S0 = sum score_inc,  sv = sum e v,  SC = sum e C,  W = sum e v v^T.
//   C  (a lane per POINT): updateDerivatives' (:511-529) gradient and Hessian terms are linear in those sums --
//         g = J^T sv,   H = J^T (SC - d2 W) J + [sv . H_ij],   J = [I | A(point)] (computePointDerivatives, :397-438)
//      -- so the Jacobian / Hessian algebra runs once per point instead of once per (point, voxel) pair: the same real numbers
//      as the reference's per-pair float evaluation, associated differently (and the Hessian's upper triangle only: H(i, j)
//      and H(j, i) are one real number whose two float evaluations in the reference differ by rounding; the control kernel mirrors it).
//   fold: the 32 columns of a wave with v_permlane32_swap / v_permlane16_swap (two columns per instruction pair: after the swap
//      one add forms lanes l + l+32 of column k in the low half and of column k+16 in the high half), then DPP row rotations --
//      170 instructions instead of 44 columns x 6 DPP steps -- then the workgroup's four waves through LDS.
// Arithmetic: per-voxel and per-point math in R (float as pclomp, double as stock PCL), a point's few terms summed in R, every
// sum beyond the point in double.  (The reference adds each float term to double sums at once; summing a point's three or so
// float terms in float first is the same order of rounding as the terms' own arithmetic, and nothing grows with the cloud.)
// A list longer than the LDS window (kCap entries: more than 16 occupied neighbours per point on average) is worked off in
// windows.  Every order is fixed by the data alone, so an evaluation is bit-reproducible, alone or in a batch.
constexpr int kNdtCols = 32;               // a row of partials: score, 6 gradient, 21 Hessian (upper triangle, row-major), pair count, 3 zeros
constexpr int kNdtOutCols = 44;            // a folded evaluation: score + 6 gradient + 36 Hessian (mirrored) + pair count

// sum of v over lanes {l, l + 32} of column a (result in lanes 0..31) and of column b (lanes 32..63)
__device__ __forceinline__ double swap32_add(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// the same across the rows of 16 lanes: rows 0 and 2 get column a (rows 0+1, 2+3), rows 1 and 3 column b
__device__ __forceinline__ double swap16_add(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// 32 columns per lane -> column sums over the wave: on return lane 16 r (r = 0..3) holds in acc[k], k = 0..7, the wave's sum of
// column k + 8 r
__device__ __forceinline__ void wave_fold32(double* acc) {
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = swap32_add(acc[k], acc[k + 16]);      // lanes 0..31: column k, lanes 32..63: column k + 16
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = swap16_add(acc[k], acc[k + 8]);        // rows 0..3: columns k, k + 8, k + 16, k + 24
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    acc[k] += dpp_row<0x128>(acc[k]);      // row_ror 8, 4, 2, 1: every lane of a row ends with the row's sum
    acc[k] += dpp_row<0x124>(acc[k]);
    acc[k] += dpp_row<0x122>(acc[k]);
    acc[k] += dpp_row<0x121>(acc[k]);
  }
}

// (pointers read from a table are generic to the compiler -- flat loads, which also wait on the LDS counter; these are global)
template <typename T> using gptr = const T __attribute__((address_space(1)))*;
template <typename T> __device__ __forceinline__ gptr<T> as_global(const T* p) { return (gptr<T>)p; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A wave's LDS: its 64 transformed points, one window of its entry list, a kept-voxel count per point
constexpr int kNdtPairCap = 1024;                   // entries per window and wave
struct NdtDerivShared {
  float4 t[kNdtDerivThreads];
  uint32_t pair[kNdtDerivThreads / 64][kNdtPairCap];            // point (6 bits) << 24 | voxel slot (24 bits)
  uint32_t cnt[kNdtDerivThreads];
  double red[kNdtDerivThreads / 64][kNdtCols];
};

struct VoxRaw { u32x4 a, b, c; };          // the first three 16-byte quarters of an NdtVoxel: mean[0..1] | mean[2], icov[0..1] | icov[2..5]  (the fourth: centroid, n)
__device__ __forceinline__ VoxRaw load_voxel_raw(const NdtVoxel* base, uint32_t slot) {
  const gptr<u32x4> q = (gptr<u32x4>)(base + slot);
  VoxRaw v;
  v.a = q[0]; v.b = q[1]; v.c = q[2];
  return v;
}

// inclusive prefix sum over the wave (DPP: inside rows of 16 lanes, then the row totals carried across)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr 1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr 2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr 4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr 8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast 15 into rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast 31 into rows 2, 3
  return v;
}

// The granule assignment depends on the slot's own point count only (never on the batch a launch carries), so an evaluation's
// sums have one order.
template <typename R, bool HESS>
__device__ __forceinline__ void ndt_derivatives_body(const NdtDev& d, const NdtPose& P, double* __restrict__ row_out, NdtDerivShared& sh) {
  constexpr bool kDouble = sizeof(R) == 8;
  constexpr int kCap = kNdtPairCap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* const s_t = sh.t + 64 * wave;
  uint32_t* const s_pair = sh.pair[wave];
  uint32_t* const s_cnt = sh.cnt + 64 * wave;
  const gptr<u32x2> words = (gptr<u32x2>)d.words;
  const gptr<double> icovd = as_global(d.icovd);
  const NdtGridInfo* g = d.info;
  const int div0 = g->div_b[0], div1 = g->div_b[1], div2 = g->div_b[2], wx = g->wx;
  const float inv = g->inv;
  const float mb0 = (float)g->min_b[0], mb1 = (float)g->min_b[1], mb2 = (float)g->min_b[2];
  const R gd2 = kDouble ? (R)P.d2d : (R)P.d2;
  double accd[kNdtCols];
#pragma unroll
  for (int k = 0; k < kNdtCols; ++k) accd[k] = 0.0;
#ifdef NDT_TIMING
  unsigned long long tm[6];
  tm[0] = tm[1] = tm[2] = tm[3] = tm[4] = wall_clock64();
  const unsigned long long cyc0 = __builtin_readcyclecounter();
#define NDT_STAMP(k) tm[k] = wall_clock64()
#else
#define NDT_STAMP(k)
#endif
  const int nwaves_own = (d.ns + 63) >> 6;                   // waves the slot's own source needs
  const int gw = blockIdx.x * (kNdtDerivThreads / 64) + wave;
  if (gw < nwaves_own) {
    // ---- phase A: the point
    const int i = (lane & 15) + 16 * (gw + (lane >> 4) * nwaves_own);
    bool live = i < d.ns;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) { const f32x4 v = ((gptr<f32x4>)d.src)[i]; s = make_float4(v.x, v.y, v.z, v.w); }
    live = live && isfinite(s.x) && isfinite(s.y) && isfinite(s.z);
    // pcl::transformPointCloud with the float final_transformation_
    const float tx = P.T[0] * s.x + P.T[1] * s.y + P.T[2] * s.z + P.T[3];
    const float ty = P.T[4] * s.x + P.T[5] * s.y + P.T[6] * s.z + P.T[7];
    const float tz = P.T[8] * s.x + P.T[9] * s.y + P.T[10] * s.z + P.T[11];
    int c0 = 0, c1 = 0, c2 = 0;
    if (live) {
      c0 = (int)(floorf(tx * inv) - mb0); c1 = (int)(floorf(ty * inv) - mb1); c2 = (int)(floorf(tz * inv) - mb2);
      // (far outside the box every row test below fails; keep the row arithmetic inside int range)
      live = !g->status && c0 >= -1 && c0 <= div0 && c1 >= -1 && c1 <= div1 && c2 >= -1 && c2 <= div2;
    }
    s_t[lane] = make_float4(tx, ty, tz, 0.f);
    s_cnt[lane] = 0u;
    const int xa = min(max(c0 - 1, 0), div0 - 1), xb = min(max(c0 + 1, 0), div0 - 1);
    uint32_t off = 0, total = 0;
    // the point's sums over its voxels
    R S0 = 0, sv0 = 0, sv1 = 0, sv2 = 0, SCxx = 0, SCxy = 0, SCxz = 0, SCyy = 0, SCyz = 0, SCzz = 0, Wxx = 0, Wxy = 0, Wxz = 0, Wyy = 0, Wyz = 0, Wzz = 0;
    uint32_t npairs_lane = 0;
    const double tdx = (double)tx, tdy = (double)ty, tdz = (double)tz;
    NDT_STAMP(1);
    for (uint32_t wbase = 0; wbase == 0 || wbase < total; wbase += kCap) {
      // the occupancy words of the nine rows (re-read for every further window of a crowded wave: L2 hits)
      uint2 wa[9], wb[9];
      uint32_t rowok = 0;
      int c1w = c1, c2w = c2;
      asm volatile("" : "+v"(c1w), "+v"(c2w));            // (keeps the 18 row addresses out of registers across phase B)
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int z = c2w + r / 3 - 1, y = c1w + r % 3 - 1;
        const bool ok = live && z >= 0 && z < div2 && y >= 0 && y < div1;
        rowok |= ok ? 1u << r : 0u;
        const int rowbase = ok ? (z * div1 + y) * wx : 0;
        { const u32x2 v = words[rowbase + (xa >> 5)]; wa[r] = make_uint2(v.x, v.y); }
        { const u32x2 v = words[rowbase + (xb >> 5)]; wb[r] = make_uint2(v.x, v.y); }
      }
      if (wbase == 0) {
        uint32_t nnb = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int x = c0 + dx;
            const uint2 wd = (x >> 5) == (xa >> 5) ? wa[r] : wb[r];
            nnb += (((rowok >> r) & 1u) && x >= 0 && x < div0 && ((wd.x >> (x & 31)) & 1u)) ? 1u : 0u;
          }
        NDT_STAMP(2);
        const uint32_t incl = wave_incl_scan_u32(nnb);
        off = incl - nnb;
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (the wave's earlier reads of its window are done)
      uint32_t k = off - wbase;                            // position in this window (wraps below it: the unsigned compare drops those)
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = c0 + dx;
          const uint2 wd = (x >> 5) == (xa >> 5) ? wa[r] : wb[r];
          if (((rowok >> r) & 1u) && x >= 0 && x < div0 && ((wd.x >> (x & 31)) & 1u)) {
            if (k < (uint32_t)kCap) s_pair[k] = ((uint32_t)lane << 24) | (wd.y + __popc(wd.x & ((1u << (x & 31)) - 1u)));
            ++k;
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the window is written: LDS is in order inside a wave
      NDT_STAMP(3);
      // ---- phase B1: a lane per entry; keep what radiusSearch returns, compacting the window in place and counting per point
      const uint32_t nwin = min((uint32_t)kCap, total - wbase);
      uint32_t kept = 0;                                   // wave-uniform
      for (uint32_t q0 = 0; q0 < nwin; q0 += 256) {
        uint32_t prs[4];
        u32x4 cen[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t q = q0 + 64 * u + lane;
          prs[u] = 0; cen[u] = u32x4{0, 0, 0, 0};
          if (q < nwin) { prs[u] = s_pair[q]; cen[u] = ((gptr<u32x4>)(d.vox + (prs[u] & 0xffffffu)))[3]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t q = q0 + 64 * u + lane;
          const float4 tp = s_t[prs[u] >> 24];
          const int vn_pts = (int)cen[u].w;
          const float ex = tp.x - __uint_as_float(cen[u].x), ey = tp.y - __uint_as_float(cen[u].y), ez = tp.z - __uint_as_float(cen[u].z);
          const bool pass = q < nwin && !(vn_pts >= 0 && vn_pts < d.min_points) && !(ex * ex + ey * ey + ez * ez > P.res2);
          const unsigned long long mask = __ballot(pass);
          const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          if (pass) { s_pair[pos] = prs[u]; atomicAdd(&s_cnt[prs[u] >> 24], 1u); }
          kept += (uint32_t)__popcll(mask);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- phase B2: a lane per point over its kept voxels (they follow one another in the compacted window); the next record is
      // on its way while this one is worked on
      const uint32_t mine = s_cnt[lane];
      s_cnt[lane] = 0u;
      const uint32_t first = wave_incl_scan_u32(mine) - mine;
      npairs_lane += mine;
      VoxRaw vn;
      vn.a = vn.b = vn.c = u32x4{0, 0, 0, 0};
      uint32_t slot_next = 0;
      if (mine) { slot_next = s_pair[first] & 0xffffffu; vn = load_voxel_raw(d.vox, slot_next); }
      for (uint32_t jn = 0; jn < mine; ++jn) {
        const VoxRaw vr = vn;
        const uint32_t slot = slot_next;
        if (jn + 1 < mine) { slot_next = s_pair[first + jn + 1] & 0xffffffu; vn = load_voxel_raw(d.vox, slot_next); }
        // x_trans - mean in double, then R (:253, :490)
        const R u0 = (R)(tdx - __hiloint2double((int)vr.a.y, (int)vr.a.x));
        const R u1 = (R)(tdy - __hiloint2double((int)vr.a.w, (int)vr.a.z));
        const R u2 = (R)(tdz - __hiloint2double((int)vr.b.y, (int)vr.b.x));
        R cxx, cxy, cxz, cyy, cyz, czz;
        if (kDouble) {
          const gptr<double> ic = icovd + (size_t)slot * 6;
          cxx = (R)ic[0]; cxy = (R)ic[1]; cxz = (R)ic[2]; cyy = (R)ic[3]; cyz = (R)ic[4]; czz = (R)ic[5];
        } else {
          cxx = __uint_as_float(vr.b.z); cxy = __uint_as_float(vr.b.w); cxz = __uint_as_float(vr.c.x);
          cyy = __uint_as_float(vr.c.y); cyz = __uint_as_float(vr.c.z); czz = __uint_as_float(vr.c.w);
        }
        // x_trans4 * c_inv4
        const R v0 = u0 * cxx + u1 * cxy + u2 * cxz;
        const R v1 = u0 * cxy + u1 * cyy + u2 * cyz;
        const R v2 = u0 * cxz + u1 * cyz + u2 * czz;
        const R qq = u0 * v0 + u1 * v1 + u2 * v2;
        const R e0 = kDouble ? (R)exp(-(double)gd2 * (double)qq / 2) : (R)expf(-(float)gd2 * (float)qq * 0.5f);   // :497
        const R e1 = gd2 * e0;                                     // :501
        // :504-505: a term whose e_x_cov_x is out of range (or NaN) adds nothing, not even its score -- here by a factor of zero
        const bool good = !(e1 > (R)1 || e1 < (R)0 || e1 != e1);
        S0 += good ? (R)(-P.d1d * (double)e0) : (R)0;                              // :499
        const R e = good ? (R)(P.d1d * (double)e1) : (R)0;                          // :508
        const R ev0 = e * v0, ev1 = e * v1, ev2 = e * v2;
        sv0 += ev0; sv1 += ev1; sv2 += ev2;
        SCxx += e * cxx; SCxy += e * cxy; SCxz += e * cxz; SCyy += e * cyy; SCyz += e * cyz; SCzz += e * czz;
        if (HESS) { Wxx += ev0 * v0; Wxy += ev0 * v1; Wxz += ev0 * v2; Wyy += ev1 * v1; Wyz += ev1 * v2; Wzz += ev2 * v2; }
      }
    }
    NDT_STAMP(4);
    // ---- phase C: the point's terms.  J = [I | a3 a4 a5], a3 = (0, pg0, pg1), a4 = (pg2, pg3, pg4), a5 = (pg5, pg6, pg7) (:397-412)
    R pg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      pg[r] = kDouble ? (R)(P.j_angd[r][0] * (double)s.x + P.j_angd[r][1] * (double)s.y + P.j_angd[r][2] * (double)s.z)
                      : (R)(P.j_ang[r][0] * s.x + P.j_ang[r][1] * s.y + P.j_ang[r][2] * s.z);
    accd[0] = (double)S0;
    // score_gradient += e (x' C) J, :511-513
    accd[1] = (double)sv0; accd[2] = (double)sv1; accd[3] = (double)sv2;
    accd[4] = (double)(sv1 * pg[0] + sv2 * pg[1]);
    accd[5] = (double)(sv0 * pg[2] + sv1 * pg[3] + sv2 * pg[4]);
    accd[6] = (double)(sv0 * pg[5] + sv1 * pg[6] + sv2 * pg[7]);
    if (HESS) {
      R ph[15];
#pragma unroll
      for (int r = 0; r < 15; ++r)                                                                             // :416
        ph[r] = kDouble ? (R)(P.h_angd[r][0] * (double)s.x + P.h_angd[r][1] * (double)s.y + P.h_angd[r][2] * (double)s.z)
                        : (R)(P.h_ang[r][0] * s.x + P.h_ang[r][1] * s.y + P.h_ang[r][2] * s.z);
      // hessian(i, j) += e (-d2 (x'CJ_i)(x'CJ_j) + x'C H_ij + J_j C J_i) summed over the point's voxels = J^T M J + [sv . H_ij],
      // M = SC - d2 W (:527-529)
      const R Mxx = SCxx - gd2 * Wxx, Mxy = SCxy - gd2 * Wxy, Mxz = SCxz - gd2 * Wxz, Myy = SCyy - gd2 * Wyy, Myz = SCyz - gd2 * Wyz, Mzz = SCzz - gd2 * Wzz;
      // M a3, M a4, M a5
      const R k30 = Mxy * pg[0] + Mxz * pg[1], k31 = Myy * pg[0] + Myz * pg[1], k32 = Myz * pg[0] + Mzz * pg[1];
      const R k40 = Mxx * pg[2] + Mxy * pg[3] + Mxz * pg[4], k41 = Mxy * pg[2] + Myy * pg[3] + Myz * pg[4], k42 = Mxz * pg[2] + Myz * pg[3] + Mzz * pg[4];
      const R k50 = Mxx * pg[5] + Mxy * pg[6] + Mxz * pg[7], k51 = Mxy * pg[5] + Myy * pg[6] + Myz * pg[7], k52 = Mxz * pg[5] + Myz * pg[6] + Mzz * pg[7];
      // sv . (a b c / b d e / c e f), a = (0, ph0, ph1) b = (0, ph2, ph3) c = (0, ph4, ph5) d = (ph6, ph7, ph8) e = (ph9, ph10, ph11) f = (ph12, ph13, ph14) (:418-437)
      const R ha = sv1 * ph[0] + sv2 * ph[1], hb = sv1 * ph[2] + sv2 * ph[3], hc = sv1 * ph[4] + sv2 * ph[5];
      const R hd = sv0 * ph[6] + sv1 * ph[7] + sv2 * ph[8], he = sv0 * ph[9] + sv1 * ph[10] + sv2 * ph[11];
      const R hf = sv0 * ph[12] + sv1 * ph[13] + sv2 * ph[14];
      accd[7] = (double)Mxx; accd[8] = (double)Mxy; accd[9] = (double)Mxz; accd[10] = (double)k30; accd[11] = (double)k40; accd[12] = (double)k50;
      accd[13] = (double)Myy; accd[14] = (double)Myz; accd[15] = (double)k31; accd[16] = (double)k41; accd[17] = (double)k51;
      accd[18] = (double)Mzz; accd[19] = (double)k32; accd[20] = (double)k42; accd[21] = (double)k52;
      accd[22] = (double)(pg[0] * k31 + pg[1] * k32 + ha);
      accd[23] = (double)(pg[2] * k30 + pg[3] * k31 + pg[4] * k32 + hb);
      accd[24] = (double)(pg[5] * k30 + pg[6] * k31 + pg[7] * k32 + hc);
      accd[25] = (double)(pg[2] * k40 + pg[3] * k41 + pg[4] * k42 + hd);
      accd[26] = (double)(pg[5] * k40 + pg[6] * k41 + pg[7] * k42 + he);
      accd[27] = (double)(pg[5] * k50 + pg[6] * k51 + pg[7] * k52 + hf);
    }
    accd[28] = (double)npairs_lane;
  }
  // ---- fold: columns over the wave, then over the workgroup's four waves (the only point where its waves meet)
  wave_fold32(accd);
  double (*const s_red)[kNdtCols] = sh.red;
  if ((lane & 15) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[wave][k + 8 * (lane >> 4)] = accd[k];
  }
  __syncthreads();
  if (threadIdx.x < kNdtCols) row_out[threadIdx.x] = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
#ifdef NDT_TIMING
  // tuning build only: the workgroup's phase stamps (10 ns ticks) in the row's three unused columns
  __syncthreads();
  if (threadIdx.x == 0) {
    tm[5] = wall_clock64();
    unsigned long long* w = reinterpret_cast<unsigned long long*>(row_out);
    w[29] = tm[0];
    w[30] = ((tm[1] - tm[0]) & 0xffff) | (((tm[2] - tm[0]) & 0xffff) << 16) | (((tm[3] - tm[0]) & 0xffff) << 32) | (((tm[4] - tm[0]) & 0xffff) << 48);
    w[31] = (tm[5] - tm[0]) | ((__builtin_readcyclecounter() - cyc0) << 32);
  }
#endif
}

// ------------------------------------------------------------------------------------------
// The Newton / More-Thuente driver on the device
// ------------------------------------------------------------------------------------------
// computeTransformation (ndt_omp_impl.hpp:81-171) and computeStepLengthMT (:757-916) are sequential 6-vector code around
// computeDerivatives calls.  Each Align is a state machine (NdtCtl) that asks for one evaluation at a time: INIT (:119) -> per
// Newton iteration TRIAL (:809-813) -> MT (the line-search loop's evaluations, :870-878) -> HESS (:912-913, only after a line
// search that looped) -> next iteration.  One round = ndt_derivatives_ctl (every running job's evaluation, grid.y = job) +
// ndt_ctl_step (a workgroup per job: fold of the job's rows in a fixed order, the 6x6 solve, the line-search decision, the next
// pose with its angular derivative tables).  The host only enqueues rounds; it never sees a score or a gradient.
//
// Provenance note: update_interval_mt / trial_value_mt / the ndt_ctl_* transitions follow pclomp/ndt_omp_impl.hpp:633-916 branch
// for branch (same variable roles: a_l, f_l, g_l, a_u, phi_0, d_psi_t, open_interval ...), including the reference's quirk that
// `interval_converged` is computed from the UN-updated interval: the iteration and derivative-call counts the parity tests
// assert depend on that control flow.  Nothing in it is data-parallel; everything it drives is this repository's own code.
struct NdtCtlOpts {
  double step_size, trans_eps;       // computeStepLengthMT's step_max; transformation_epsilon_
  double d1, d2;                     // gauss_d1_, gauss_d2_ (:86-93)
  float res2;                        // resolution^2
  int32_t max_iterations;
  int32_t double_math;               // stock pcl::NormalDistributionsTransform arithmetic (NdtWithGicp): fill the double tables
  int32_t pad_;
};

enum NdtPhase : int32_t { kNdtInit = 0, kNdtTrial = 1, kNdtMt = 2, kNdtHess = 3, kNdtDone = 4, kNdtEvalOnly = 5 };

struct NdtCtl {
  int32_t phase, slot, it, deriv_calls;
  int32_t interval_converged, open_interval, step_iterations, done_round;
  double p[6], x_t[6], dir[6], eval_p[6];
  double sc, g[6], H[36], last_pairs;
  // computeStepLengthMT's locals
  double phi_0, d_phi_0, a_t, a_l, a_u, f_l, g_l, f_u, g_u, phi_t, d_phi_t, psi_t, d_psi_t, step_max, step_min;
  float Tf[16];                      // the pose matrix of the last evaluation asked for = final_transformation_ at the end
  NdtPose pose;                      // the evaluation wanted next
};

struct NdtResult {                   // written to page-locked host memory by the step that ends a job
  float Tf[16];
  double sc, last_pairs;
  int32_t it, deriv_calls, done_round, pad_;
};

constexpr double kMtMu = 1.e-4, kMtNu = 0.9;

__host__ __device__ inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }     // ndt_omp.h auxilaryFunction_PsiMT
__host__ __device__ inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

__host__ __device__ inline bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {   // :633-670
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  else if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  else if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

__host__ __device__ inline double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {   // :674-753
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double nxt = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? fmin(a_t + 0.66 * (a_u - a_t), nxt) : fmax(a_t + 0.66 * (a_u - a_t), nxt);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

// Eigen::JacobiSVD<Matrix6d>(H, FullU | FullV).solve(b) (:127-129) = the pseudo-inverse applied to b.  For a matrix of full
// numerical rank that is H^-1 b, formed here by Gaussian elimination with partial pivoting (a few hundred dependent
// instructions on one lane; a Jacobi SVD is several thousand, on the critical path of every round).  When the pivots span more
// than ten decades the one-sided Jacobi SVD below decides which directions count, with JacobiSVD's threshold.
__host__ __device__ __noinline__ void svd_solve6(const double* H /*row-major*/, const double* b, double* x) {
  double A[36], V[36];
  for (int i = 0; i < 36; ++i) { A[i] = H[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int k = 0; k < 6; ++k) { app += A[6 * k + p] * A[6 * k + p]; aqq += A[6 * k + q] * A[6 * k + q]; apq += A[6 * k + p] * A[6 * k + q]; }
        if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-16 * sqrt(app * aqq)) continue;
        rotated = true;
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; ++k) {
          const double ap = A[6 * k + p], aq = A[6 * k + q];
          A[6 * k + p] = c * ap - s * aq; A[6 * k + q] = s * ap + c * aq;
          const double vp = V[6 * k + p], vq = V[6 * k + q];
          V[6 * k + p] = c * vp - s * vq; V[6 * k + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sig[6], smax = 0;
  for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += A[6 * k + j] * A[6 * k + j]; sig[j] = sqrt(s); smax = fmax(smax, sig[j]); }
  const double thr = 2.220446049250313e-16 * 6 * smax;
  for (int i = 0; i < 6; ++i) x[i] = 0;
  for (int j = 0; j < 6; ++j) {
    if (!(sig[j] > thr)) continue;
    double ub = 0;                                   // u_j . b with u_j = A[:, j] / sig_j
    for (int k = 0; k < 6; ++k) ub += A[6 * k + j] * b[k];
    ub /= sig[j] * sig[j];
    for (int i = 0; i < 6; ++i) x[i] += V[6 * i + j] * ub;
  }
}

__host__ __device__ inline void ndt_solve6(const double* H /*row-major*/, const double* b, double* x) {
  double A[6][7];
  for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) A[i][j] = H[6 * i + j]; A[i][6] = b[i]; }
  double pmin = 1.7976931348623157e308, pmax = 0;
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    double best = fabs(A[c][c]);
#pragma unroll
    for (int r = c + 1; r < 6; ++r) { const double v = fabs(A[r][c]); if (v > best) { best = v; piv = r; } }
    if (!(best > 0) || !(best < 1.7976931348623157e308)) { ok = false; break; }
    pmin = fmin(pmin, best); pmax = fmax(pmax, best);
#pragma unroll
    for (int r = c + 1; r < 6; ++r)
      if (r == piv) {
#pragma unroll
        for (int j = 0; j < 7; ++j) { const double t = A[c][j]; A[c][j] = A[r][j]; A[r][j] = t; }
      }
    const double ip = 1.0 / A[c][c];
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double f = A[r][c] * ip;
#pragma unroll
      for (int j = c + 1; j < 7; ++j) A[r][j] -= f * A[c][j];
    }
  }
  if (!ok || !(pmin > 1e-10 * pmax)) { svd_solve6(H, b, x); return; }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double sacc = A[r][6];
#pragma unroll
    for (int j = r + 1; j < 6; ++j) sacc -= A[r][j] * x[j];
    x[r] = sacc / A[r][r];
  }
}

// ---- the transitions (one lane).  They end by asking for an evaluation at j.eval_p (phase kNdtTrial / kNdtMt: a new pose matrix
// is due; kNdtHess: the same pose once more, Hessian only) or with kNdtDone.
__host__ __device__ inline void ndt_ctl_request(NdtCtl& j, const double* x, bool hess, int32_t ph) {
  for (int i = 0; i < 6; ++i) j.eval_p[i] = x[i];
  j.pose.compute_hessian = hess ? 1 : 0;
  j.phase = ph;
}

__host__ __device__ __noinline__ void ndt_ctl_newton(NdtCtl& j, const NdtCtlOpts& o);

__host__ __device__ inline void ndt_ctl_finish_iteration(NdtCtl& j, const NdtCtlOpts& o) {
  const double dp_norm = j.a_t;                                                    // :142
  for (int i = 0; i < 6; ++i) j.p[i] += j.dir[i] * dp_norm;                        // :143, :152
  const bool converged = j.it > o.max_iterations || (j.it && fabs(dp_norm) < o.trans_eps);   // :158-162
  j.it++;                                                                          // :164
  if (converged) { j.phase = kNdtDone; return; }
  ndt_ctl_newton(j, o);
}

// the line-search loop's head (:867): another trial value, the closing Hessian, or the end of the iteration
__host__ __device__ inline void ndt_ctl_mt_continue(NdtCtl& j, const NdtCtlOpts& o) {
  if (!j.interval_converged && j.step_iterations < 10 && !(j.psi_t <= 0 && j.d_phi_t <= -kMtNu * j.d_phi_0)) {
    j.a_t = j.open_interval ? trial_value_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.psi_t, j.d_psi_t)
                            : trial_value_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.phi_t, j.d_phi_t);
    j.a_t = fmax(fmin(j.a_t, j.step_max), j.step_min);
    for (int i = 0; i < 6; ++i) j.x_t[i] = j.p[i] + j.dir[i] * j.a_t;
    ndt_ctl_request(j, j.x_t, false, kNdtMt);
    return;
  }
  if (j.step_iterations) { ndt_ctl_request(j, j.x_t, true, kNdtHess); return; }       // :912-913
  ndt_ctl_finish_iteration(j, o);
}

// one Newton iteration up to its first evaluation (:121-141, :757-813)
__host__ __device__ __noinline__ void ndt_ctl_newton(NdtCtl& j, const NdtCtlOpts& o) {
  for (;;) {
    double mg[6], dp[6];
    for (int i = 0; i < 6; ++i) mg[i] = -j.g[i];
    ndt_solve6(j.H, mg, dp);                                                           // :127-129
    double dp_norm = 0;
    for (int i = 0; i < 6; ++i) dp_norm += dp[i] * dp[i];
    dp_norm = sqrt(dp_norm);
    if (dp_norm == 0 || dp_norm != dp_norm) { j.phase = kNdtDone; return; }        // :134-139
    for (int i = 0; i < 6; ++i) j.dir[i] = dp[i] / dp_norm;                        // :141
    // ---- computeStepLengthMT(p, dir, dp_norm, step_size, trans_eps / 2, ...) :757-916
    const double step_init = dp_norm;
    j.step_max = o.step_size; j.step_min = o.trans_eps / 2;
    j.phi_0 = -j.sc;
    j.d_phi_0 = 0;
    for (int i = 0; i < 6; ++i) j.d_phi_0 -= j.g[i] * j.dir[i];
    j.a_t = 0;
    bool skip = false;
    if (j.d_phi_0 >= 0) {
      if (j.d_phi_0 == 0) skip = true;
      else { j.d_phi_0 *= -1; for (int i = 0; i < 6; ++i) j.dir[i] = -j.dir[i]; }
    }
    if (!skip) {
      j.a_l = 0; j.a_u = 0;
      j.f_l = psi_mt(j.a_l, j.phi_0, j.phi_0, j.d_phi_0, kMtMu); j.g_l = dpsi_mt(j.d_phi_0, j.d_phi_0, kMtMu);
      j.f_u = j.f_l; j.g_u = j.g_l;
      j.interval_converged = (j.step_max - j.step_min) > 0 ? 1 : 0;                // :795 (sic: the loop never runs with the wrapper's settings)
      j.open_interval = 1;
      j.step_iterations = 0;
      j.a_t = fmax(fmin(step_init, j.step_max), j.step_min);
      for (int i = 0; i < 6; ++i) j.x_t[i] = j.p[i] + j.dir[i] * j.a_t;
      ndt_ctl_request(j, j.x_t, true, kNdtTrial);                                  // :803-813
      return;
    }
    // d_phi_0 == 0: no step; the iteration ends where it began
    const bool converged = j.it > o.max_iterations || (j.it && fabs(j.a_t) < o.trans_eps);
    j.it++;
    if (converged) { j.phase = kNdtDone; return; }
  }
}

// the evaluation the job asked for has come back: out = score, gradient, hessian (36), pair count
__host__ __device__ inline void ndt_ctl_result(NdtCtl& j, const double* out, const NdtCtlOpts& o) {
  j.deriv_calls++;
  j.last_pairs = out[43];
  switch (j.phase) {
    case kNdtInit:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];
      ndt_ctl_newton(j, o);
      break;
    case kNdtTrial:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];
      j.phi_t = -j.sc; j.d_phi_t = 0;
      for (int i = 0; i < 6; ++i) j.d_phi_t -= j.g[i] * j.dir[i];
      j.psi_t = psi_mt(j.a_t, j.phi_t, j.phi_0, j.d_phi_0, kMtMu); j.d_psi_t = dpsi_mt(j.d_phi_t, j.d_phi_0, kMtMu);
      ndt_ctl_mt_continue(j, o);
      break;
    case kNdtMt:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];                             // (no Hessian in the loop, :872)
      j.phi_t = -j.sc; j.d_phi_t = 0;
      for (int i = 0; i < 6; ++i) j.d_phi_t -= j.g[i] * j.dir[i];
      j.psi_t = psi_mt(j.a_t, j.phi_t, j.phi_0, j.d_phi_0, kMtMu); j.d_psi_t = dpsi_mt(j.d_phi_t, j.d_phi_0, kMtMu);
      if (j.open_interval && (j.psi_t <= 0 && j.d_psi_t >= 0)) {
        j.open_interval = 0;
        j.f_l = j.f_l + j.phi_0 - kMtMu * j.d_phi_0 * j.a_l; j.g_l = j.g_l + kMtMu * j.d_phi_0;
        j.f_u = j.f_u + j.phi_0 - kMtMu * j.d_phi_0 * j.a_u; j.g_u = j.g_u + kMtMu * j.d_phi_0;
      }
      j.interval_converged = (j.open_interval ? update_interval_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.psi_t, j.d_psi_t)
                                              : update_interval_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.phi_t, j.d_phi_t)) ? 1 : 0;
      j.step_iterations++;
      ndt_ctl_mt_continue(j, o);
      break;
    case kNdtHess:
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];                            // the Hessian alone: score and gradient stay the loop's last
      ndt_ctl_finish_iteration(j, o);
      break;
    default:
      break;
  }
}

// computeAngleDerivatives (:288-393) from the sines / cosines of p[3..5] (with the reference's small-angle rule applied by the caller)
__host__ __device__ inline void ndt_angle_tables(double cx, double sx, double cy, double sy, double cz, double sz, bool dbl, NdtPose& P) {
  const double j[8][3] = {
      {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)}, {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
      {(-sy * cz), sy * sz, cy}, {sx * cy * cz, (-sx * cy * sz), sx * sy}, {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
      {(-cy * sz), (-cy * cz), 0}, {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0}, {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
  const double hh[15][3] = {
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
      {(cx * cy * cz), (-cx * cy * sz), (cx * sy)}, {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
      {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0}, {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
      {(-cy * cz), (cy * sz), (sy)}, {(-sx * sy * cz), (sx * sy * sz), (sx * cy)}, {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
      {(sy * sz), (sy * cz), 0}, {(-sx * cy * sz), (-sx * cy * cz), 0}, {(cx * cy * sz), (cx * cy * cz), 0},
      {(-cy * cz), (cy * sz), 0}, {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
  if (dbl) {
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) P.j_angd[r][c] = j[r][c];
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) P.h_angd[r][c] = hh[r][c];
  } else {
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) P.j_ang[r][c] = (float)j[r][c];
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) P.h_ang[r][c] = (float)hh[r][c];
  }
}

// Translation(p0..2) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z), all float (:146-149, :803-806), from the float
// cosines / sines of the float-rounded angles
__host__ __device__ inline void ndt_pose_matrix_f32(const double* p, float ca, float sa, float cb, float sb, float cc, float sc, float* T /*row-major 4x4*/) {
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  const float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  const float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float M[9], R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += Rx[3 * i + k] * Ry[3 * k + j]; M[3 * i + j] = s; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += M[3 * i + k] * Rz[3 * k + j]; R[3 * i + j] = s; }
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = (float)p[i]; }
  T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

// grid = (workgroups per source, jobs): job blockIdx.y evaluates its pose against the table of its slot; a finished job's
// workgroups leave at once
template <typename R>
__global__ __launch_bounds__(kNdtDerivThreads, sizeof(R) == 4 ? 4 : 3) void ndt_derivatives_ctl(const NdtDev* __restrict__ devs, const NdtCtl* __restrict__ ctl) {
  const NdtCtl& c = ctl[blockIdx.y];
  if (c.phase == kNdtDone) return;
  const NdtDev d = devs[c.slot];
  __shared__ NdtDerivShared sh;
  if (c.pose.compute_hessian) ndt_derivatives_body<R, true>(d, c.pose, d.partials + (size_t)blockIdx.x * kNdtCols, sh);
  else ndt_derivatives_body<R, false>(d, c.pose, d.partials + (size_t)blockIdx.x * kNdtCols, sh);
}

// A workgroup per job: the fold of the job's rows (16 thread groups take every 16th row, then one thread per column adds the group
// sums: a fixed order, so an evaluation is bit-reproducible whatever else the launch holds), the state machine's step, the next
// pose.  flags[job] (page-locked) = (round + 1) << 8 | phase tells the host how far the job is; a job that ends leaves its result
// in res[job] (page-locked) and the final transformation, as doubles, in the pair input block of its slot -- where the fitness
// pass (pcl::Registration::getFitnessScore, ndt.cc:60) takes its pose from.
constexpr int kNdtStepThreads = 512;
__global__ __launch_bounds__(kNdtStepThreads) void ndt_ctl_step(const NdtDev* __restrict__ devs, NdtCtl* __restrict__ ctl, int nblocks, const NdtCtlOpts o, int round,
                                                      PairInput* __restrict__ in, uint32_t* __restrict__ flags, NdtResult* __restrict__ res, double* __restrict__ out_host) {
  static_assert(sizeof(NdtCtl) % 8 == 0, "NdtCtl is copied as 8-byte words");
  NdtCtl* const jg = ctl + blockIdx.x;
  if (jg->phase == kNdtDone) return;
  // the job's state in LDS while the step runs: the sequential code below indexes its vectors in loops (private arrays would go
  // to scratch memory)
  __shared__ NdtCtl j;
  constexpr int kGroups = kNdtStepThreads / 32;
  __shared__ double s_g[kGroups][kNdtCols];
  __shared__ double s_out[kNdtOutCols];
  __shared__ double s_trig[12];
  __shared__ int s_new_pose;
  {
    const unsigned long long* srcw = reinterpret_cast<const unsigned long long*>(jg);
    unsigned long long* dstw = reinterpret_cast<unsigned long long*>(&j);
    for (int k = threadIdx.x; k < (int)(sizeof(NdtCtl) / 8); k += blockDim.x) dstw[k] = srcw[k];
  }
  const NdtDev d = devs[jg->slot];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  {
    double t = 0;
    int k = grp;
    for (; k + 7 * kGroups < nblocks; k += 8 * kGroups) {            // eight loads in flight; added in row order
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = d.partials[(size_t)(k + u * kGroups) * kNdtCols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += a[u];
    }
    for (; k < nblocks; k += kGroups) t += d.partials[(size_t)k * kNdtCols + c];
    s_g[grp][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < kNdtCols) {
    double t = 0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) t += s_g[g][threadIdx.x];
    s_g[0][threadIdx.x] = t;                               // (thread c reads column c of every group before it overwrites group 0's)
  }
  __syncthreads();
  if (threadIdx.x < kNdtOutCols) {
    // score, gradient, the Hessian mirrored from its upper triangle, pair count
    const int k = threadIdx.x;
    double v;
    if (k < 7) v = s_g[0][k];
    else if (k == 43) v = s_g[0][28];
    else {
      int a = (k - 7) / 6, b = (k - 7) % 6;
      if (a > b) { const int t = a; a = b; b = t; }
      v = s_g[0][7 + a * 6 - a * (a - 1) / 2 + (b - a)];
    }
    s_out[k] = v;
    d.out[k] = v;
    if (out_host) out_host[(size_t)blockIdx.x * kNdtOutCols + k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (j.phase == kNdtEvalOnly) { j.deriv_calls++; j.last_pairs = s_out[43]; j.phase = kNdtDone; }
    else ndt_ctl_result(j, s_out, o);
    s_new_pose = j.phase != kNdtDone;
  }
  __syncthreads();
  if (s_new_pose) {
    // the cosines / sines of the next pose: lanes 0..2 those of the float-rounded angles (the float pose matrix), lanes 3..5
    // those of the angles themselves (the angular derivative tables, with the reference's |angle| < 1e-4 rule, :290-324)
    if (threadIdx.x < 6) {
      const double ang = j.eval_p[3 + threadIdx.x % 3];
      double sn, cs;
      if (threadIdx.x < 3) { sincos((double)(float)ang, &sn, &cs); sn = (double)(float)sn; cs = (double)(float)cs; }
      else if (fabs(ang) < 10e-5) { sn = 0.0; cs = 1.0; }
      else sincos(ang, &sn, &cs);
      s_trig[2 * threadIdx.x] = cs; s_trig[2 * threadIdx.x + 1] = sn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (j.phase == kNdtTrial || j.phase == kNdtMt)
        ndt_pose_matrix_f32(j.eval_p, (float)s_trig[0], (float)s_trig[1], (float)s_trig[2], (float)s_trig[3], (float)s_trig[4], (float)s_trig[5], j.Tf);
      for (int i = 0; i < 12; ++i) j.pose.T[i] = j.Tf[i];
    }
    if (threadIdx.x == 64) ndt_angle_tables(s_trig[6], s_trig[7], s_trig[8], s_trig[9], s_trig[10], s_trig[11], o.double_math != 0, j.pose);
  } else if (threadIdx.x == 0) {
    j.done_round = round;
    NdtResult& r = res[blockIdx.x];
    for (int i = 0; i < 16; ++i) r.Tf[i] = j.Tf[i];
    r.sc = j.sc; r.last_pairs = j.last_pairs; r.it = j.it; r.deriv_calls = j.deriv_calls; r.done_round = round;
    PairInput& pi = in[j.slot];
    for (int i = 0; i < 16; ++i) pi.guess[i] = (double)j.Tf[i];      // getFinalTransformation().cast<double>() (ndt.cc:61), row-major
  }
  __syncthreads();
  {
    const unsigned long long* srcw = reinterpret_cast<const unsigned long long*>(&j);
    unsigned long long* dstw = reinterpret_cast<unsigned long long*>(jg);
    for (int k = threadIdx.x; k < (int)(sizeof(NdtCtl) / 8); k += blockDim.x) dstw[k] = srcw[k];
  }
  if (threadIdx.x == 0) {
    __threadfence_system();
    flags[blockIdx.x] = ((uint32_t)(round + 1) << 8) | (uint32_t)j.phase;
  }
}

// ------------------------------------------------------------------------------------------
// fitness score: exact 1-NN distances over the table's own points
// ------------------------------------------------------------------------------------------
// pcl::Registration::getFitnessScore (ndt.cc:60, ndt_gicp.cc:88,101): the source moved by the float final transformation, the
// squared distance of every point to its nearest target point, their mean.  The target's points already lie sorted by voxel, mid
// cell and fine cell, so the table IS the search structure -- no second grid is built over the raw target.
//   ndt_fit_near  a lane per query: the 3 x 3 x 3 fine cells (1/16 voxel) around it, 27 independent hash probes (nine in flight at
//                 a time), then the runs of the cells that exist.  A point outside that cube is at least a fine cell away, so a best
//                 distance below that is the answer -- for most of a scan that lies on its submap.
//   ndt_fit_mid   sixteen lanes per query the first pass could not settle: shells of mid cells (1/4 voxel) around the query's own, a
//                 lane per cell (hash probe, skipped when the cell's box is farther than the best so far), its run walked by that
//                 lane; after shell R everything unseen is at least R quarter-voxels + (distance to the own mid cell's wall) away.
//   ndt_fit_far   a wave per query with nothing within kNdtFitMidRings quarter-voxels (a scan's far points, ahead of what the submap
//                 covers): over the occupied voxels' boxes -- the smallest farthest-corner distance bounds the answer from above
//                 (every listed voxel holds a point), then only the voxels whose nearest corner is within that bound are walked.
// The lattice is exact in the scaled coordinate fl(p * inv) (floor and the fractional part of a float are exact operations); `marg`
// covers that one rounding on both sides.  Distances: float, as FLANN's L2 functor.
constexpr int kNdtFitMidRings = 3;          // shells of mid cells the second pass walks: settles everything within 3/4 voxel
constexpr int kNdtFitMidBlocks = 8192;
constexpr int kNdtFitFarBlocks = 4096;

struct NdtQuery { float t[3]; int iv[3]; int c16[3]; float f[3]; bool finite; };
__device__ __forceinline__ NdtQuery ndt_fit_query(const NdtDev& d, const PairInput& in, int i) {
  const NdtGridInfo* g = d.info;
  const float4 s = d.src[i];
  NdtQuery q;
  q.finite = isfinite(s.x) && isfinite(s.y) && isfinite(s.z);
  // transformPointCloud(*input_, input_transformed, final_transformation_): the float 4x4
  const double* G = in.guess;
  q.t[0] = (float)G[0] * s.x + (float)G[1] * s.y + (float)G[2] * s.z + (float)G[3];
  q.t[1] = (float)G[4] * s.x + (float)G[5] * s.y + (float)G[6] * s.z + (float)G[7];
  q.t[2] = (float)G[8] * s.x + (float)G[9] * s.y + (float)G[10] * s.z + (float)G[11];
  q.finite = q.finite && isfinite(q.t[0]) && isfinite(q.t[1]) && isfinite(q.t[2]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float ps = q.t[c] * g->inv, fl = floorf(ps);
    // (clamped far outside the box: every voxel test fails there anyway, and the int arithmetic stays in range)
    q.iv[c] = (int)fminf(fmaxf(fl - (float)g->min_b[c], -1.0e6f), 1.0e6f);
    q.f[c] = ps - fl;
    q.c16[c] = min(15, (int)(q.f[c] * 16.0f));
  }
  return q;
}
__device__ __forceinline__ float ndt_d2(const float4 p, const float* t) {
  const float dx = p.x - t[0], dy = p.y - t[1], dz = p.z - t[2];
  return dx * dx + dy * dy + dz * dz;
}
// one lane over a run of points: four loads in flight (a lane that waits for every point in turn pays a memory latency per point;
// the clamped repeats at the run's end change no minimum)
__device__ __forceinline__ float ndt_scan_run(const float4* __restrict__ vpts, uint32_t j0, uint32_t j1, const float* t, float best) {
  for (uint32_t j = j0; j < j1; j += 4) {
    const uint32_t last = j1 - 1u;
    const float4 p0 = vpts[j], p1 = vpts[min(j + 1u, last)], p2 = vpts[min(j + 2u, last)], p3 = vpts[min(j + 3u, last)];
    best = fminf(fminf(best, ndt_d2(p0, t)), fminf(ndt_d2(p1, t), fminf(ndt_d2(p2, t), ndt_d2(p3, t))));
  }
  return best;
}

__global__ __launch_bounds__(256) void ndt_fit_near(const NdtDev* __restrict__ devs, const PairInput* __restrict__ in, int first_slot) {
  const int slot = first_slot + blockIdx.y;
  const NdtDev d = devs[slot];
  NdtGridInfo* g = d.info;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = i < d.ns;
  __shared__ uint4 s_ent[8][256];
  NdtQuery q{};
  if (have) q = ndt_fit_query(d, in[slot], i);
  float best = INFINITY;
  if (have && q.finite && !g->status) {
    const int div0 = g->div_b[0], div1 = g->div_b[1], div2 = g->div_b[2], wx = g->wx;
    // the cube in fine coordinates, and the (at most 2 x 2 x 2) mid cells it touches: their first probes go out together
    int gf[3], mlo[3], mhi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { gf[c] = q.iv[c] * 16 + q.c16[c]; mlo[c] = (gf[c] - 1) >> 2; mhi[c] = (gf[c] + 1) >> 2; }
    const uint32_t maskc = (1u << d.log2cells) - 1u;
    const gptr<u32x4> cells = (gptr<u32x4>)d.cells;          // an entry = two 16-byte quarters: key, start, end | fmask
    unsigned long long key[8];
    uint32_t hp[8];
    u32x4 e0[8], e1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int mz = mlo[2] + (c >> 2), my = mlo[1] + ((c >> 1) & 1), mx = mlo[0] + (c & 1);
      const int vz = mz >> 2, vy = my >> 2, vx = mx >> 2;
      const bool want = mz <= mhi[2] && my <= mhi[1] && mx <= mhi[0] && vz >= 0 && vz < div2 && vy >= 0 && vy < div1 && vx >= 0 && vx < div0;
      const uint32_t code = ((uint32_t)((vz * div1 + vy) * wx + (vx >> 5)) << 5) | (uint32_t)(vx & 31);
      key[c] = want ? ndt_mid_key(code, mz & 3, my & 3, mx & 3) : ~0ull;
      hp[c] = ndt_cell_hash(key[c], d.log2cells);
      e0[c] = cells[2 * hp[c]]; e1[c] = cells[2 * hp[c] + 1];
    }
    // the entries found, parked in LDS: the cube's rows below pick theirs by index
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint4 found = make_uint4(0u, 0u, 0u, 0u);              // start, end, fine mask: an absent cell has an empty mask
      if (key[c] != ~0ull) {
        u32x4 a0 = e0[c], a1 = e1[c];
        uint32_t hpos = hp[c];
        for (;;) {                                           // (linear probing: the first probe is the entry nearly always)
          const unsigned long long kk = ((unsigned long long)a0.y << 32) | a0.x;
          if (kk == key[c] || kk == ~0ull) break;
          hpos = (hpos + 1u) & maskc;
          a0 = cells[2 * hpos]; a1 = cells[2 * hpos + 1];
        }
        if ((((unsigned long long)a0.y << 32) | a0.x) == key[c]) found = make_uint4(a0.z, a0.w, a1.x, a1.y);
      }
      s_ent[c][threadIdx.x] = found;
    }
    // the cube's nine (z, y) rows of fine cells, each cut at most once by a mid cell's wall along x: a piece inside one mid cell is
    // one run of points.  All bounds first (their loads go out together), then the points.
    uint32_t r0[18], r1[18];
#pragma unroll
    for (int r = 0; r < 18; ++r) {
      const int fz = gf[2] + (r / 6) - 1, fy = gf[1] + ((r / 2) % 3) - 1, xs = r & 1;
      const int mz = fz >> 2, my = fy >> 2, mx = mlo[0] + xs;
      r0[r] = 0u; r1[r] = 0u;
      if (mx <= mhi[0]) {
        const uint4 en = s_ent[((mz - mlo[2]) << 2) | ((my - mlo[1]) << 1) | xs][threadIdx.x];
        const unsigned long long fm = ((unsigned long long)en.w << 32) | en.z;
        const int ax = max(gf[0] - 1, 4 * mx) - 4 * mx, bx = min(gf[0] + 1, 4 * mx + 3) - 4 * mx;
        const int row = ((fz & 3) * 4 + (fy & 3)) * 4;
        const uint32_t lo = (uint32_t)__popcll(fm & ((1ull << (row + ax)) - 1ull));
        const uint32_t hi = (uint32_t)__popcll(fm & ((2ull << (row + bx)) - 1ull));
        if (hi != lo) { r0[r] = d.fpos[en.x + lo]; r1[r] = hi == (uint32_t)__popcll(fm) ? en.y : d.fpos[en.x + hi]; }
      }
    }
#pragma unroll
    for (int r = 0; r < 18; ++r) best = ndt_scan_run(d.vpts, r0[r], r1[r], q.t, best);
  }
  if (have) d.fit_d2[i] = best;
  // anything outside the cube is at least a fine cell away
  const float bound = (0.0625f - g->marg) * g->res * 0.999999f;
  const bool open = have && q.finite && !(best <= bound * bound);
  // the open queries of the workgroup take one stretch of the list (one returning atomic per workgroup: one per query queued up
  // on the counter for 18 ns each)
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_base;
  uint32_t total;
  const uint32_t off = block_excl_scan(open ? 1u : 0u, s_w, &total);
  if (threadIdx.x == 0) s_base = total ? atomicAdd(&g->nlist, total) : 0u;
  __syncthreads();
  if (open) d.qlist[s_base + off] = (uint32_t)i;
}

template <int CTRL>
__device__ __forceinline__ float dpp_rowf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float row_min16(float v) {        // min over the lane's row of 16, in every lane of the row
  v = fminf(v, dpp_rowf<0x128>(v));       // row_ror 8, 4, 2, 1
  v = fminf(v, dpp_rowf<0x124>(v));
  v = fminf(v, dpp_rowf<0x122>(v));
  v = fminf(v, dpp_rowf<0x121>(v));
  return v;
}

__global__ __launch_bounds__(256) void ndt_fit_mid(const NdtDev* __restrict__ devs, const PairInput* __restrict__ in, int first_slot) {
  const int slot = first_slot + blockIdx.y;
  const NdtDev d = devs[slot];
  NdtGridInfo* g = d.info;
  const int l16 = threadIdx.x & 15;
  const uint32_t nlist = g->nlist;
  const int div0 = g->div_b[0], div1 = g->div_b[1], div2 = g->div_b[2];
  const uint32_t rows = gridDim.x * (blockDim.x >> 4);
  // (whole rows loop together: the row-wide reductions need every lane of a row in the loop)
  for (uint32_t k = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); k < nlist; k += rows) {
    const int i = (int)d.qlist[k];
    const NdtQuery q = ndt_fit_query(d, in[slot], i);
    float best = d.fit_d2[i];                                // what the fine cube held (or infinity)
    bool settled = false;
    if (!g->status) {
      int gm[3];
      float fm[3];                                           // position inside the own mid cell, in voxels: [0, 0.25)
#pragma unroll
      for (int c = 0; c < 3; ++c) { gm[c] = q.iv[c] * 4 + (q.c16[c] >> 2); fm[c] = q.f[c] - 0.25f * (float)(q.c16[c] >> 2); }
      const float wall = fminf(fminf(fminf(fm[0], 0.25f - fm[0]), fminf(fm[1], 0.25f - fm[1])), fminf(fm[2], 0.25f - fm[2]));
      for (int R = 1; R <= kNdtFitMidRings && !settled; ++R) {
        const int side = 2 * R + 1, cells = side * side * side;
        for (int c = l16; c < cells; c += 16) {
          const int dx = c % side - R, dy = (c / side) % side - R, dz = c / (side * side) - R;
          if (R > 1 && max(abs(dx), max(abs(dy), abs(dz))) < R) continue;      // seen in an earlier shell (shell 1 is the whole cube)
          const int m0 = gm[0] + dx, m1 = gm[1] + dy, m2 = gm[2] + dz;
          const int vx = m0 >> 2, vy = m1 >> 2, vz = m2 >> 2;
          if (vx < 0 || vx >= div0 || vy < 0 || vy >= div1 || vz < 0 || vz >= div2) continue;
          // distance from the query to the cell's box, in voxels
          const float gx = dx > 0 ? 0.25f * (float)dx - fm[0] : (dx < 0 ? fm[0] - 0.25f * (float)(dx + 1) : 0.f);
          const float gy = dy > 0 ? 0.25f * (float)dy - fm[1] : (dy < 0 ? fm[1] - 0.25f * (float)(dy + 1) : 0.f);
          const float gz = dz > 0 ? 0.25f * (float)dz - fm[2] : (dz < 0 ? fm[2] - 0.25f * (float)(dz + 1) : 0.f);
          const float gap = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz) - g->marg, 0.f) * g->res * 0.999999f;
          if (gap * gap > best) continue;
          const uint32_t code = ((uint32_t)((vz * div1 + vy) * g->wx + (vx >> 5)) << 5) | (uint32_t)(vx & 31);
          const NdtCell* e = ndt_cell_find(d, ndt_mid_key(code, m2 & 3, m1 & 3, m0 & 3));
          if (!e) continue;
          best = ndt_scan_run(d.vpts, e->start, e->end, q.t, best);
        }
        best = row_min16(best);
        const float bound = fmaxf(0.25f * (float)R + wall - g->marg, 0.f) * g->res * 0.999999f;
        settled = best <= bound * bound;
      }
    }
    if (l16 == 0) {
      d.fit_d2[i] = best;
      if (!settled) d.qleft[atomicAdd(&g->nleft, 1u)] = (uint32_t)i;
    }
  }
}

__global__ __launch_bounds__(256) void ndt_fit_far(const NdtDev* __restrict__ devs, const PairInput* __restrict__ in, int first_slot) {
  const int slot = first_slot + blockIdx.y;
  const NdtDev d = devs[slot];
  const NdtGridInfo* g = d.info;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nleft = g->nleft;
  const int nocc = g->nocc;
  __shared__ float s_m[4];
  // a workgroup per query (there are few of them, and each looks at every occupied voxel's box twice)
  for (uint32_t k = blockIdx.x; k < nleft; k += gridDim.x) {
    const int i = (int)d.qleft[k];
    const NdtQuery q = ndt_fit_query(d, in[slot], i);
    const float qs0 = q.t[0] * g->inv, qs1 = q.t[1] * g->inv, qs2 = q.t[2] * g->inv;     // the query in scaled coordinates
    // an upper bound in voxels: the best so far, and the farthest corner of the nearest boxes
    float up = sqrtf(d.fit_d2[i]) / g->res * 1.000001f + g->marg;
    for (int v = threadIdx.x; v < nocc; v += blockDim.x) {
      const float4 c = d.vbox[v];
      const float fx = fmaxf(fabsf(qs0 - c.x), fabsf(qs0 - c.x - 1.f)), fy = fmaxf(fabsf(qs1 - c.y), fabsf(qs1 - c.y - 1.f)), fz = fmaxf(fabsf(qs2 - c.z), fabsf(qs2 - c.z - 1.f));
      up = fminf(up, sqrtf(fx * fx + fy * fy + fz * fz) + g->marg);
    }
    up = wave_min(up);
    if (lane == 0) s_m[wave] = up;
    __syncthreads();
    up = fminf(fminf(s_m[0], s_m[1]), fminf(s_m[2], s_m[3]));
    __syncthreads();
    float best = INFINITY;
    for (int v0 = 64 * wave; v0 < nocc; v0 += 256) {       // a wave takes 64 voxels at a time and walks the near ones' points together
      const int v = v0 + lane;
      bool near = false;
      if (v < nocc) {
        const float4 c = d.vbox[v];
        const float nx = fmaxf(fmaxf(c.x - qs0, qs0 - c.x - 1.f), 0.f), ny = fmaxf(fmaxf(c.y - qs1, qs1 - c.y - 1.f), 0.f), nz = fmaxf(fmaxf(c.z - qs2, qs2 - c.z - 1.f), 0.f);
        near = sqrtf(nx * nx + ny * ny + nz * nz) - g->marg <= up;
      }
      unsigned long long live = __ballot(near);
      while (live) {
        const int l = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        const uint32_t a = d.vstart[v0 + l], b = d.vstart[v0 + l + 1];
        for (uint32_t j = a + lane; j < b; j += 64) best = fminf(best, ndt_d2(d.vpts[j], q.t));
      }
    }
    best = wave_min(best);
    if (lane == 0) s_m[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) d.fit_d2[i] = fminf(fminf(fminf(s_m[0], s_m[1]), fminf(s_m[2], s_m[3])), d.fit_d2[i]);
    __syncthreads();
  }
}

// mean of the squared NN distances (pcl::Registration::getFitnessScore, ndt.cc:60): grid = (64, slots); the partial sums go
// straight into page-locked host memory (row first_slot-relative): no copy
__global__ __launch_bounds__(256) void fitness_partial(const NdtDev* __restrict__ devs, int first_slot, double* partials_all) {
  const NdtDev d = devs[first_slot + blockIdx.y];
  const float* d2 = d.fit_d2;
  const int n = d.ns;
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

// the fitness search's counters of slots [first, first + n) back to zero (every search starts from empty lists)
__global__ void ndt_fit_reset(const NdtDev* __restrict__ devs, int first_slot, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) { NdtGridInfo* g = devs[first_slot + k].info; g->nlist = 0u; g->nleft = 0u; }
}

}  // namespace smhip
