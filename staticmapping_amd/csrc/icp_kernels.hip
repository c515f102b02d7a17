// icp_kernels.hip -- hand-written gfx950 kernels for the IcpFast hot path.
//
// Kernel inventory (SURVEY.md §2b ids in brackets; reference lines are in /root/reference):
//   tgt_reduce / grid_setup / grid_mark / grid_rank / grid_count / grid_cscan / grid_scatter_idx / grid_place
//        target centring (icp_fast.cc:457-463) + the search structure that replaces the
//        libnabo kd-tree rebuilt on every Align (icp_fast.cc:464-467)
//   nn_grid, nn_brute   [K1]  ApplyTransform + FindClosests      (icp_fast.cc:486-493, 169-180)
//   accumulate          [K2/K3] GetDistsQuantile bin + ErrorElements + point-to-plane sums
//                                                               (icp_fast.cc:65-90, 100-166, 256-303)
//   finalize            [K2/K4] exact quantile inside the boundary bin, 6x6 solve, SE(3) update,
//                       CheckConvergence, score                  (icp_fast.cc:204-254, 306-323, 377-405, 513-523)
//
// Everything here is HBM/L2-bound gather, scan and reduction work: wave64 shuffles + LDS, no MFMA.
#include <type_traits>
#include "smhip_device.h"
#include "kd_median_tree.h"

namespace smhip {

// XCD-aware work mapping for the heavy per-point kernels.  They are launched as a 1-D grid of
// nblk * 8 * ceil(npairs / 8) workgroups; the dispatcher deals consecutive workgroup ids round-robin
// over the 8 XCDs (observed, MI355X_MICROARCH.md), so XCD k = id % 8 walks the pairs k, k + 8, ... one
// after the other and every block of a pair runs on the same XCD: the pair's grid words, sorted
// target and match arrays stay in that XCD's 4 MiB L2 instead of being spread over all eight.
// Only speed depends on the placement, never correctness.
__device__ __forceinline__ bool xcd_block(int nblk, int npairs, int b_pair_base, int& pair, int& blk) {
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int local = (j / nblk) * 8 + (id & 7);
  pair = b_pair_base + local;
  blk = j % nblk;
  return local < npairs;
}

// ------------------------------------------------------------------------------------------
// target preparation
// ------------------------------------------------------------------------------------------
// Partial sums (f64) and bbox (f32, exact) of the raw target; grid = (kTgtReduceBlocks, pairs).
__global__ __launch_bounds__(256) void tgt_reduce(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const int nt = b.in[pair].nt;
  const float4* tp = b.tgt_p + (size_t)pair * b.nt_cap;
  double sx = 0, sy = 0, sz = 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int chunk = (nt + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * chunk, hi = min(nt, lo + chunk);
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    float4 p = tp[j];
    sx += p.x; sy += p.y; sz += p.z;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ double s_sum[4][3];
  __shared__ float s_mn[4][3], s_mx[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if (lane == 0) {
    s_sum[wave][0] = sx; s_sum[wave][1] = sy; s_sum[wave][2] = sz;
    for (int d = 0; d < 3; ++d) { s_mn[wave][d] = mn[d]; s_mx[wave][d] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* out = b.tpart + ((size_t)pair * kTgtReduceBlocks + blockIdx.x) * 16;
    for (int d = 0; d < 3; ++d) {
      out[d] = s_sum[0][d] + s_sum[1][d] + s_sum[2][d] + s_sum[3][d];
      out[3 + d] = fminf(fminf(s_mn[0][d], s_mn[1][d]), fminf(s_mn[2][d], s_mn[3][d]));
      out[6 + d] = fmaxf(fmaxf(s_mx[0][d], s_mx[1][d]), fmaxf(s_mx[2][d], s_mx[3][d]));
    }
  }
}

__device__ __forceinline__ void mat4_mul_rm(const double* a, const double* c, double* out) {
  double r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * c[4 * k + j];
      r[4 * i + j] = s;
    }
  for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// One thread per pair: mean, grid geometry, G = T(-mu) * guess, loop state reset.
// Per-Align scratch of the first npairs slots zeroed in one launch (four memsets cost a single pair ~30 us of launch gaps).
__global__ __launch_bounds__(256) void reset_scratch(IcpDev b, int first, int npairs) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  uint4* bits4 = reinterpret_cast<uint4*>(b.bits) + (size_t)kMaxGridWords / 4 * first;   // hipMalloc alignment; kMaxGridWords % 4 == 0
  uint32_t* cc = b.ccount + (size_t)(b.nt_cap + 1) * first;
  uint32_t* hh = b.hist + (size_t)kHistBins * first;
  for (size_t k = tid; k < (size_t)kMaxGridWords / 4 * npairs; k += nth) bits4[k] = make_uint4(0, 0, 0, 0);
  for (size_t k = tid; k < (size_t)(b.nt_cap + 1) * npairs; k += nth) cc[k] = 0;
  for (size_t k = tid; k < (size_t)kHistBins * npairs; k += nth) hh[k] = 0;
  for (size_t k = tid; k < (size_t)kSearchHist * npairs; k += nth) b.search_hist[(size_t)kSearchHist * first + k] = 0;
  if (tid == 0) *b.done_count = 0;
  for (size_t k = tid; k < (size_t)kOneSyncWords * kOnePairs; k += nth) b.one_sync[k] = 0;
  for (size_t k = tid; k < (size_t)kHistBins * kOnePairs; k += nth) b.one_hist[k] = 0;
  for (size_t k = tid; k < sizeof(PairState) / 4 * kOnePairs; k += nth) reinterpret_cast<uint32_t*>(b.one_ctr)[k] = 0;
}

// The per-Align part of grid_setup alone (pose chain, loop state), for a pair whose target -- and with it the mean, the
// grid geometry and the sorted target -- is unchanged since the last build: the front end aligns many scans against one
// key frame (map_builder.cc:379-392), the back end many candidates against one submap.
__global__ void pose_setup(IcpDev b, int npairs) {
  const int pair = b.pair_base + blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= b.pair_base + npairs) return;
  PairState* st = &b.state[pair];
  const PairInput* in = &b.in[pair];
  st->ns = in->ns; st->has_normals = in->has_normals;
  for (int i = 0; i < 16; ++i) st->guess[i] = in->guess[i];
  if (st->grid_invalid) {                 // the kept "structure" is grid_setup's placeholder for a non-finite target: fail again
    for (int i = 0; i < 16; ++i) st->result[(i % 4) * 4 + i / 4] = st->guess[i];
    st->iter = 0; st->score = 0; st->score_mismatch = 0; st->kept = 0; st->searched_total = 0; st->fallback_total = 0; st->hard_total = 0; st->refine_total = 0;
    st->status = 1; st->done = 1;
    atomicAdd(b.done_count, 1u);
    return;
  }
  const double Tmi[16] = {1, 0, 0, -st->mu[0], 0, 1, 0, -st->mu[1], 0, 0, 1, -st->mu[2], 0, 0, 0, 1};
  mat4_mul_rm(Tmi, st->guess, st->G);
  for (int i = 0; i < 16; ++i) st->T_iter[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 12; ++i) { st->M[i] = st->G[i]; st->M_prev[i] = st->G[i]; }
  st->quat[0][0] = 1; st->quat[0][1] = st->quat[0][2] = st->quat[0][3] = 0;
  st->trans[0][0] = st->trans[0][1] = st->trans[0][2] = 0;
  st->n_hist = 1;
  st->iter = 0; st->done = 0; st->status = 0;
  st->unresolved_count = 0; st->fallback_ticket = 0; st->fallback_total = 0; st->hard_count = 0; st->hard_total = 0;
  st->min_lb_key = 0xffffffffu; st->refine = 0; st->refine_total = 0; st->deferred_count = 0; st->searched_total = 0;
  for (int c = 0; c < 4; ++c) st->nabo_count[c] = 0;
  st->pot_a = 0; st->pot_b = 0; st->step_a = 0; st->step_b = 0;
  st->rcap2 = b.ball_radius * b.ball_radius;
  st->kept = 0; st->limit_key = 0; st->score = 0; st->score_mismatch = 0;
  st->band_lo = 0; st->band_hi = -1; st->spec_ok = 0; st->spec_hits = 0;
}
// per-Align scratch that is not part of the search structure: histogram + finished-pairs counter
__global__ __launch_bounds__(256) void reset_scratch_light(IcpDev b, int first, int npairs) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  uint32_t* hh = b.hist + (size_t)kHistBins * first;
  for (size_t k = tid; k < (size_t)kHistBins * npairs; k += nth) hh[k] = 0;
  for (size_t k = tid; k < (size_t)kSearchHist * npairs; k += nth) b.search_hist[(size_t)kSearchHist * first + k] = 0;
  if (tid == 0) *b.done_count = 0;
  for (size_t k = tid; k < (size_t)kOneSyncWords * kOnePairs; k += nth) b.one_sync[k] = 0;
  for (size_t k = tid; k < (size_t)kHistBins * kOnePairs; k += nth) b.one_hist[k] = 0;
  for (size_t k = tid; k < sizeof(PairState) / 4 * kOnePairs; k += nth) reinterpret_cast<uint32_t*>(b.one_ctr)[k] = 0;
}

__global__ void grid_setup(IcpDev b, int npairs) {
  const int pair = b.pair_base + blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= b.pair_base + npairs) return;
  PairState* st = &b.state[pair];
  const double* part = b.tpart + (size_t)pair * kTgtReduceBlocks * 16;
  double s[3] = {0, 0, 0};
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < kTgtReduceBlocks; ++k)
    for (int d = 0; d < 3; ++d) {
      s[d] += part[16 * k + d];
      mn[d] = fminf(mn[d], (float)part[16 * k + 3 + d]);
      mx[d] = fmaxf(mx[d], (float)part[16 * k + 6 + d]);
    }
  const PairInput* in = &b.in[pair];
  const int nt = in->nt;
  st->ns = in->ns; st->nt = nt; st->has_normals = in->has_normals;
  for (int i = 0; i < 16; ++i) st->guess[i] = in->guess[i];
  double mu[3];
  for (int d = 0; d < 3; ++d) { mu[d] = s[d] / nt; st->mu[d] = mu[d]; }        // icp_fast.cc:457-458
  // centred bbox, the same rounding the mark/scatter kernels apply to every point
  float cmin[3], cmax[3];
  for (int d = 0; d < 3; ++d) { cmin[d] = (float)((double)mn[d] - mu[d]); cmax[d] = (float)((double)mx[d] - mu[d]); }
  // A NaN / Inf target coordinate would make the cell counts below overflow (and grid_mark write outside `bits`): such a
  // pair fails cleanly instead -- one cell, nothing searched, status = invalid argument, counted as finished.
  bool finite_box = true;
  for (int d = 0; d < 3; ++d) finite_box = finite_box && isfinite(cmin[d]) && isfinite(cmax[d]) && isfinite(mu[d]) && cmax[d] >= cmin[d];
  if (!finite_box) {
    st->h = 1.0f; st->inv_h = 1.0f;
    for (int d = 0; d < 3; ++d) st->origin[d] = 0.f;
    st->nx = st->ny = st->nz = 1; st->wx = 1; st->nw = 1; st->nocc = 0;
    for (int i = 0; i < 16; ++i) { st->T_iter[i] = (i % 5 == 0) ? 1.0 : 0.0; st->G[i] = st->guess[i]; st->result[(i % 4) * 4 + i / 4] = st->guess[i]; }
    for (int i = 0; i < 12; ++i) { st->M[i] = st->guess[i]; st->M_prev[i] = st->guess[i]; }
    st->n_hist = 1; st->iter = 0; st->score = 0; st->score_mismatch = 0; st->kept = 0; st->limit_key = 0;
    st->unresolved_count = 0; st->fallback_ticket = 0; st->fallback_total = 0; st->hard_count = 0; st->hard_total = 0;
    st->min_lb_key = 0xffffffffu; st->refine = 0; st->refine_total = 0; st->deferred_count = 0; st->searched_total = 0;
  for (int c = 0; c < 4; ++c) st->nabo_count[c] = 0;
  st->pot_a = 0; st->pot_b = 0; st->step_a = 0; st->step_b = 0;
    st->rcap2 = 0.f;
    st->band_lo = 0; st->band_hi = -1; st->spec_ok = 0; st->spec_hits = 0;
    st->status = 1;                        // SMHIP_ERR_INVALID_ARGUMENT
    st->done = 1;
    st->grid_invalid = 1;
    atomicAdd(b.done_count, 1u);
    return;
  }
  st->grid_invalid = 0;
  float h = b.grid_cell;
  int nx, ny, nz, wx;
  for (;;) {
    const float inv = 1.0f / h;
    nx = (int)floorf((cmax[0] - cmin[0]) * inv) + 2;
    ny = (int)floorf((cmax[1] - cmin[1]) * inv) + 2;
    nz = (int)floorf((cmax[2] - cmin[2]) * inv) + 2;
    wx = (nx + 31) >> 5;
    const double words = (double)wx * ny * nz;
    if (words <= (double)kMaxGridWords && nx < 32768 && ny < 32768 && nz < 32768) break;
    h *= 1.25992105f;   // 2^(1/3): halve the cell count
  }
  st->h = h; st->inv_h = 1.0f / h;
  for (int d = 0; d < 3; ++d) st->origin[d] = cmin[d] - 0.5f * h;
  st->nx = nx; st->ny = ny; st->nz = nz; st->wx = wx; st->nw = wx * ny * nz;
  // G = T(-mu) * guess   (icp_fast.cc:469), T_iter = I (:473)
  double Tmi[16] = {1, 0, 0, -mu[0], 0, 1, 0, -mu[1], 0, 0, 1, -mu[2], 0, 0, 0, 1};
  mat4_mul_rm(Tmi, st->guess, st->G);
  for (int i = 0; i < 16; ++i) st->T_iter[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 12; ++i) { st->M[i] = st->G[i]; st->M_prev[i] = st->G[i]; }
  st->quat[0][0] = 1; st->quat[0][1] = st->quat[0][2] = st->quat[0][3] = 0;      // :478-479
  st->trans[0][0] = st->trans[0][1] = st->trans[0][2] = 0;
  st->n_hist = 1;
  st->iter = 0; st->done = 0; st->status = 0;
  st->unresolved_count = 0; st->fallback_ticket = 0; st->fallback_total = 0; st->hard_count = 0; st->hard_total = 0;
  st->min_lb_key = 0xffffffffu; st->refine = 0; st->refine_total = 0; st->deferred_count = 0; st->searched_total = 0;
  for (int c = 0; c < 4; ++c) st->nabo_count[c] = 0;
  st->pot_a = 0; st->pot_b = 0; st->step_a = 0; st->step_b = 0;
  st->rcap2 = b.ball_radius * b.ball_radius;
  st->kept = 0; st->limit_key = 0; st->score = 0; st->score_mismatch = 0; st->nocc = 0;
  st->band_lo = 0; st->band_hi = -1; st->spec_ok = 0; st->spec_hits = 0;
}

__device__ __forceinline__ float3 centre_point(const float4 p, const double* mu) {
  return make_float3((float)((double)p.x - mu[0]), (float)((double)p.y - mu[1]), (float)((double)p.z - mu[2]));
}

__global__ __launch_bounds__(256) void grid_mark(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= st->nt) return;
  const float3 c = centre_point(b.tgt_p[(size_t)pair * b.nt_cap + j], st->mu);
  int ix = (int)floorf((c.x - st->origin[0]) * st->inv_h);
  int iy = (int)floorf((c.y - st->origin[1]) * st->inv_h);
  int iz = (int)floorf((c.z - st->origin[2]) * st->inv_h);
  ix = min(max(ix, 0), st->nx - 1); iy = min(max(iy, 0), st->ny - 1); iz = min(max(iz, 0), st->nz - 1);
  const uint32_t w = (uint32_t)((iz * st->ny + iy) * st->wx + (ix >> 5));
  const uint32_t bit = ix & 31;
  atomicOr(&b.bits[(size_t)pair * kMaxGridWords + w], 1u << bit);
  b.tcell[(size_t)pair * b.nt_cap + j] = (w << 5) | bit;
}

// words[w] = {bits, exclusive popcount rank}; nocc.  Tiles of 4096 words, 16-byte coalesced loads, running carry between
// tiles.  grid = (segments, pairs): a batch gives every pair one 1024-thread workgroup; a launch with few pairs (one scan
// pair, the 500 k-point NDT / GICP targets) cuts the pair's words into segments, and a segment's workgroup first
// popcounts the words before its own (reads only, no scan) to get its starting rank -- 66 -> 14 us for one pair.
__global__ __launch_bounds__(1024) void grid_rank(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  const int nw = st->nw;
  const uint32_t* bits = b.bits + (size_t)pair * kMaxGridWords;
  uint2* words = b.words + (size_t)pair * kMaxGridWords;
  __shared__ uint32_t s_w[17];
  const int tiles = (nw + 4095) / 4096;
  const int tiles_per_seg = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * tiles_per_seg * 4096;
  const int hi = min(nw, lo + tiles_per_seg * 4096);
  if (lo >= nw) return;
  uint32_t carry = 0;
  if (lo > 0) {                                            // rank at the start of this segment (lo is a multiple of 4096)
    uint32_t mine = 0;
    for (int w = (int)threadIdx.x * 4; w < lo; w += 4096) {
      const uint4 v = *reinterpret_cast<const uint4*>(bits + w);
      mine += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    (void)block_excl_scan(mine, s_w, &carry);
  }
  for (int t0 = lo; t0 < hi; t0 += 4096) {
    const int w = t0 + (int)threadIdx.x * 4;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (w + 3 < nw) v = *reinterpret_cast<const uint4*>(bits + w);      // kMaxGridWords keeps this 16-B aligned
    else {
      if (w < nw) v.x = bits[w];
      if (w + 1 < nw) v.y = bits[w + 1];
      if (w + 2 < nw) v.z = bits[w + 2];
    }
    const uint32_t c0 = __popc(v.x), c1 = __popc(v.y), c2 = __popc(v.z), c3 = __popc(v.w);
    uint32_t total;
    uint32_t run = carry + block_excl_scan(c0 + c1 + c2 + c3, s_w, &total);
    if (w < nw) words[w] = make_uint2(v.x, run);
    run += c0;
    if (w + 1 < nw) words[w + 1] = make_uint2(v.y, run);
    run += c1;
    if (w + 2 < nw) words[w + 2] = make_uint2(v.z, run);
    run += c2;
    if (w + 3 < nw) words[w + 3] = make_uint2(v.w, run);
    carry += total;
  }
  if (hi >= nw && threadIdx.x == 0) st->nocc = (int)carry;   // the segment that holds the last word
}

// Row-occupancy bitmap: bit (z * ny + y) of rowbits says whether grid row (y, z) holds any point.  The ring searches walk
// cell blocks of up to (2 r + 1)^2 rows per query; against a sparse far range nearly all of them are empty, and proving a
// row empty through `words` costs two dependent loads.  One thread per 32-row word, no atomics, nothing to zero.
__global__ __launch_bounds__(256) void grid_rowbits(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int rows = st->ny * st->nz, wx = st->wx;
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= kMaxRowWords) return;
  uint32_t out = 0;
  if (w * 32 < rows) {
    const uint32_t* bits = b.bits + (size_t)pair * kMaxGridWords;
    for (int r = 0; r < 32; ++r) {
      const int row = w * 32 + r;
      if (row >= rows) break;
      uint32_t any = 0;
      for (int k = 0; k < wx; ++k) any |= bits[(size_t)row * wx + k];
      if (any) out |= 1u << r;
    }
  }
  b.rowbits[(size_t)pair * kMaxRowWords + w] = out;
}
// rows y in [y0, y1] of slab z that hold points, in ascending y: f(y)
template <typename F>
__device__ __forceinline__ void for_each_occupied_row(const uint32_t* __restrict__ rb, int ny, int z, int y0, int y1, F f) {
  for (int yb = y0; yb <= y1; yb += 32) {
    const int cnt = min(32, y1 - yb + 1);
    const int ry = z * ny + yb, w = ry >> 5, sh = ry & 31;
    const unsigned long long two = (unsigned long long)rb[w] | ((unsigned long long)rb[w + 1] << 32);
    uint32_t m = (uint32_t)(two >> sh) & (cnt == 32 ? 0xffffffffu : ((1u << cnt) - 1u));
    while (m) { const int o = __ffs((int)m) - 1; m &= m - 1u; f(yb + o); }
  }
}
__device__ __forceinline__ bool row_occupied(const uint32_t* __restrict__ rb, int ny, int z, int y) {
  const int ry = z * ny + y;
  return (rb[ry >> 5] >> (ry & 31)) & 1u;
}

__global__ __launch_bounds__(256) void grid_count(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= st->nt) return;
  const uint32_t cell = b.tcell[(size_t)pair * b.nt_cap + j];
  const uint2 wd = b.words[(size_t)pair * kMaxGridWords + (cell >> 5)];
  const uint32_t slot = wd.y + __popc(wd.x & ((1u << (cell & 31)) - 1u));
  const uint32_t ord = atomicAdd(&b.ccount[(size_t)pair * (b.nt_cap + 1) + slot], 1u);
  b.tslot[(size_t)pair * b.nt_cap + j] = slot;
  b.tord[(size_t)pair * b.nt_cap + j] = ord;
}

__global__ __launch_bounds__(1024) void grid_cscan(IcpDev b) {
  const int pair = b.pair_base + blockIdx.x;
  const PairState* st = &b.state[pair];
  const int n = st->nocc;
  const uint32_t* cnt = b.ccount + (size_t)pair * (b.nt_cap + 1);
  uint32_t* cs = b.cstart + (size_t)pair * (b.nt_cap + 1);
  __shared__ uint32_t s_w[17];
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
  uint32_t c = 0;
  for (int k = lo; k < hi; ++k) c += cnt[k];
  uint32_t total;
  uint32_t run = block_excl_scan(c, s_w, &total);
  for (int k = lo; k < hi; ++k) { cs[k] = run; run += cnt[k]; }
  if (threadIdx.x == 0) cs[n] = total;
}

// The cell-sorted target in two steps.  grid_count dealt every point a place in its cell in the order the atomics happened to
// arrive; positions must not depend on that (the tie rule of every search is "smallest sorted position").  Step 1 leaves only the
// point's INDEX at that place (perm: the tcell array, free once the points are counted); step 2 ranks every point among the
// indices of its cell's run -- a handful of 4-byte reads -- and writes the 32 bytes of point + normal once, at their final
// place: cells in linear order, a cell's points by caller index.  (Before: the 32 bytes scattered in arrival order, then one
// thread per cell insertion-sorting them in global memory -- as long as grid_mark for nothing but a few swaps.)
// (sort_cells = 0 -- the NDT fitness pass, the GICP neighbourhoods: their searches carry the tie rule on the caller index explicitly,
// so a cell's points may stand in arrival order: one scatter, no ranking.  Their targets are dense -- tens of points per cell.)
__global__ __launch_bounds__(256) void grid_scatter(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= st->nt) return;
  const size_t o = (size_t)pair * b.nt_cap;
  const float3 c = centre_point(b.tgt_p[o + j], st->mu);
  const uint32_t pos = b.cstart[(size_t)pair * (b.nt_cap + 1) + b.tslot[o + j]] + b.tord[o + j];
  b.tq[o + pos] = make_float4(c.x, c.y, c.z, __int_as_float(j));
  float4 n = b.tgt_n[o + j];
  n.w = 0.f;
  b.tn[o + pos] = n;
}
__global__ __launch_bounds__(256) void grid_scatter_idx(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= st->nt) return;
  const size_t o = (size_t)pair * b.nt_cap;
  b.tcell[o + b.cstart[(size_t)pair * (b.nt_cap + 1) + b.tslot[o + j]] + b.tord[o + j]] = (uint32_t)j;
}
__global__ __launch_bounds__(256) void grid_place(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  const PairState* st = &b.state[pair];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= st->nt) return;
  const size_t o = (size_t)pair * b.nt_cap;
  const uint32_t* cs = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const uint32_t slot = b.tslot[o + j];
  const uint32_t c0 = cs[slot], c1 = cs[slot + 1];
  const float4 p = b.tgt_p[o + j];
  float4 n = b.tgt_n[o + j];
  uint32_t rank = 0;
  for (uint32_t k = c0; k < c1; ++k) rank += b.tcell[o + k] < (uint32_t)j ? 1u : 0u;
  const float3 c = centre_point(p, st->mu);
  b.tq[o + c0 + rank] = make_float4(c.x, c.y, c.z, __int_as_float(j));
  n.w = 0.f;
  b.tn[o + c0 + rank] = n;
}

// ------------------------------------------------------------------------------------------
// K1: transform + exact 1-NN
// ------------------------------------------------------------------------------------------
// Tie rule everywhere: among equidistant targets the smallest SORTED position j wins.  Positions
// are deterministic (cells in linear order, points inside a cell ordered by original index, see
// grid_place), so brute force, tile search, ring search and fallback agree bit for bit.
struct Best { float d2; int j; float s2; };   // nearest (squared distance, position) and runner-up squared distance

__device__ __forceinline__ float dist2(const float4 t, float qx, float qy, float qz) {
  const float dx = qx - t.x, dy = qy - t.y, dz = qz - t.z;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
// candidates visited in ascending j: strict "<" keeps the smallest j among ties
__device__ __forceinline__ void test_ascending(const float4 t, int j, float qx, float qy, float qz, Best& best) {
  const float d = dist2(t, qx, qy, qz);
  if (d < best.d2) { best.d2 = d; best.j = j; }
}
// same, also tracking the runner-up distance (what the next iteration's certificate needs).  (Ordering the distances as
// unsigned integers -- they are never negative or NaN -- drops the v_max_f32 x, x canonicalisation the compiler puts before
// every float min / max, but cost nn_ball_lds 36 more bytes of scratch per lane at its 80-VGPR budget and 6 % of its speed.)
__device__ __forceinline__ void test_ascending_ru(const float4 t, int j, float qx, float qy, float qz, Best& best) {
  const float d = dist2(t, qx, qy, qz);
  // (best, runner-up) = the two smallest of {best, runner-up, d}; best <= runner-up always holds, so the new runner-up is
  // the median of the three: ONE v_med3_f32 instead of max + min with a canonicalising v_max_f32 x, x in front of each
  // (3 of the 15 vector instructions per candidate).  Distances are never NaN here (finite queries, finite targets).
  best.s2 = __builtin_amdgcn_fmed3f(d, best.d2, best.s2);
  if (d < best.d2) { best.d2 = d; best.j = j; }
}
// two candidates per step with packed fp32 (v_pk_add/mul/fma_f32): 6 instead of 9 VALU per candidate
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void test_ascending2(const float4 a, int ja, const float4 c, int jc,
                                                float qx, float qy, float qz, Best& best) {
  const f32x2 dx = (f32x2){qx, qx} - (f32x2){a.x, c.x};
  const f32x2 dy = (f32x2){qy, qy} - (f32x2){a.y, c.y};
  const f32x2 dz = (f32x2){qz, qz} - (f32x2){a.z, c.z};
  f32x2 d = dx * dx;
  d = __builtin_elementwise_fma(dy, dy, d);
  d = __builtin_elementwise_fma(dz, dz, d);
  if (d.x < best.d2) { best.d2 = d.x; best.j = ja; }
  if (d.y < best.d2) { best.d2 = d.y; best.j = jc; }
}
// arbitrary visiting order: explicit tie rule
__device__ __forceinline__ void test_any_order(const float4 t, int j, float qx, float qy, float qz, Best& best) {
  const float d = dist2(t, qx, qy, qz);
  if (d < best.d2 || (d == best.d2 && j < best.j)) { best.d2 = d; best.j = j; }
}

// source point k of the packed copy (one 12-byte load), as the float4 the kernels were written for
__device__ __forceinline__ float4 ld_src(const IcpDev& b, size_t k) {
  const float3 v = *reinterpret_cast<const float3*>(b.src3 + 3 * k);
  return make_float4(v.x, v.y, v.z, 0.f);
}
__device__ __forceinline__ float ld_lb(const IcpDev& b, size_t k) { return b.lb[k]; }
__device__ __forceinline__ void st_lb(const IcpDev& b, size_t k, float v) { b.lb[k] = v; }
// The 4-byte shadow of (match, bound): bits [14:0] the match (0x7fff: none), [31:15] the bound as a 17-bit float -- sign, 4 exponent
// bits (2^-10 .. 2^3 m; 0 = the value zero, below a millimetre: "unknown"), 12 mantissa bits TRUNCATED, so its magnitude is never
// above the recorded one (a smaller bound certifies less, never wrongly) and at most 2^-13 below it.  (An IEEE half -- 10 mantissa
// bits -- made 0.8 % more certificates fail, those of far points, whose records carry metres of motion potential and whose
// searches are the long ones: the listed search took 17 % longer.)
__device__ __forceinline__ uint32_t pack_match(int j, float lbv) {
  const uint32_t u = __float_as_uint(lbv), a0 = u & 0x7fffffffu;
  uint32_t code = 0;
  if (a0 >= 0x3a800000u) {                                   // >= 2^-10
    const uint32_t a = min(a0, 0x417ff800u);                 // < 16
    code = ((u >> 31) << 16) | (((a >> 23) - 116u) << 12) | ((a >> 11) & 0xfffu);
  }
  return (code << 15) | (uint32_t)(j < 0 || j >= 0x7fff ? 0x7fff : j);
}
__device__ __forceinline__ float shadow_bound(uint32_t m) {
  // [31] sign, [30:27] exponent - 116 (0: no bound), [26:15] the mantissa's top twelve bits: exponent and mantissa move down as one field
  const uint32_t bits = (((m & 0x7fff8000u) >> 4) + 0x3a000000u) | (m & 0x80000000u);
  return (m & 0x78000000u) ? __uint_as_float(bits) : 0.f;
}
__device__ __forceinline__ int shadow_match(uint32_t m) { const int j = (int)(m & 0x7fffu); return j == 0x7fff ? -1 : j; }
// a query's match and certificate bound: the two arrays every kernel reads, and their 4-byte shadow (IcpDev::mb)
__device__ __forceinline__ void st_match(const IcpDev& b, size_t k, int j, float lbv) {
  b.idx[k] = j;
  b.lb[k] = lbv;
  b.mb[k] = pack_match(j, lbv);
}
// certificate bounds are stored with the pair's motion potential at the time of the search added (PairState::pot_a / pot_b)
struct Pot { float a, b, sa, sb; };
__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }
__device__ __forceinline__ float pot_at(const Pot& p, float sn) { return fmaf(p.a, sn, p.b); }
__device__ __forceinline__ float with_pot(float lb, float P) { return lb > 0.f ? lb + P : (lb < 0.f ? lb - P : 0.f); }
// what a stored bound is worth now: |stored| - P(now) - rounding slack (the float roundings of the stored sum, of P and of |s|)
// (1e-4 P also covers |s| taken as |q - t| by the search lanes of nn_ball_lds for a guess whose rotation is only orthonormal to 1e-5)
__device__ __forceinline__ float bound_now(float stored, float P) { return fabsf(stored) - P - (1.0e-5f * (1.0f + fabsf(stored)) + 1.0e-4f * P); }

// src (float4 rows as uploaded) -> src3 for pairs [first, first + npairs)
__global__ __launch_bounds__(256) void pack_source(IcpDev b, int first, int npairs) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const size_t total = (size_t)npairs * (size_t)b.ns_cap;
  float* out = const_cast<float*>(b.src3);
  for (size_t k = tid; k < total; k += nth) {
    const int pair = first + (int)(k / (size_t)b.ns_cap);
    const int i = (int)(k % (size_t)b.ns_cap);
    if (i >= b.in[pair].ns) continue;
    const size_t g = (size_t)pair * b.ns_cap + i;
    const float4 v = b.src[g];
    *reinterpret_cast<float3*>(out + 3 * g) = make_float3(v.x, v.y, v.z);
  }
}

__device__ __forceinline__ void transform_point(const double* M, const float4 s, double& px, double& py, double& pz) {
  const double x = s.x, y = s.y, z = s.z;
  px = fma(M[0], x, fma(M[1], y, fma(M[2], z, M[3])));      // ApplyTransform, cloud_types.cc:288-302
  py = fma(M[4], x, fma(M[5], y, fma(M[6], z, M[7])));
  pz = fma(M[8], x, fma(M[9], y, fma(M[10], z, M[11])));
}

// wave-wide min / max of an int, uniform result: the same DPP pattern as wave_incl_scan with min / max in place of +, the
// total read from lane 63
__device__ __forceinline__ int wave_min_i(int v) {
  constexpr int kId = 0x7fffffff;
  v = min(v, dpp_or<0x111, 0xf>(kId, v));
  v = min(v, dpp_or<0x112, 0xf>(kId, v));
  v = min(v, dpp_or<0x114, 0xf>(kId, v));
  v = min(v, dpp_or<0x118, 0xf>(kId, v));
  v = min(v, dpp_or<0x142, 0xa>(kId, v));
  v = min(v, dpp_or<0x143, 0xc>(kId, v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
  constexpr int kId = -0x7fffffff - 1;
  v = max(v, dpp_or<0x111, 0xf>(kId, v));
  v = max(v, dpp_or<0x112, 0xf>(kId, v));
  v = max(v, dpp_or<0x114, 0xf>(kId, v));
  v = max(v, dpp_or<0x118, 0xf>(kId, v));
  v = max(v, dpp_or<0x142, 0xa>(kId, v));
  v = max(v, dpp_or<0x143, 0xc>(kId, v));
  return __builtin_amdgcn_readlane(v, 63);
}

// occupied-cell slots [s_begin, s_end) of the cells x in [xa, xb] of one grid row
__device__ __forceinline__ void row_slots(const uint2* __restrict__ words, int rowbase, int xa, int xb,
                                          uint32_t& s_begin, uint32_t& s_end) {
  const int w0 = xa >> 5, w1 = xb >> 5;
  const uint2 a = words[rowbase + w0];
  const uint2 c = (w1 == w0) ? a : words[rowbase + w1];
  s_begin = a.y + __popc(a.x & ((1u << (xa & 31)) - 1u));
  s_end = c.y + __popc(c.x & (0xffffffffu >> (31 - (xb & 31))));
}

// distance from q to the nearest face of the cell block [X0,X1]x[Y0,Y1]x[Z0,Z1] that still has grid
// beyond it: every target point outside the block is at least that far away.  INFINITY = the block
// covers the whole grid (exhaustive).  A small fp32 slack covers points sitting on a cell face.
__device__ __forceinline__ float block_guarantee(const PairState* st, float qx, float qy, float qz,
                                                 int X0, int X1, int Y0, int Y1, int Z0, int Z1) {
  const float h = st->h;
  float g = INFINITY;
  if (X0 > 0) g = fminf(g, qx - (st->origin[0] + (float)X0 * h));
  if (X1 < st->nx - 1) g = fminf(g, (st->origin[0] + (float)(X1 + 1) * h) - qx);
  if (Y0 > 0) g = fminf(g, qy - (st->origin[1] + (float)Y0 * h));
  if (Y1 < st->ny - 1) g = fminf(g, (st->origin[1] + (float)(Y1 + 1) * h) - qy);
  if (Z0 > 0) g = fminf(g, qz - (st->origin[2] + (float)Z0 * h));
  if (Z1 < st->nz - 1) g = fminf(g, (st->origin[2] + (float)(Z1 + 1) * h) - qz);
  return g - 1.0e-3f * h;
}

__device__ __forceinline__ int cell_coord(float q, float o, float inv_h) {
  return (int)floorf(fminf(fmaxf((q - o) * inv_h, -1.0e6f), 1.0e6f));
}

__device__ __forceinline__ void search_row(const uint2* __restrict__ words, const uint32_t* __restrict__ cstart,
                                           const float4* __restrict__ tq, int rowbase, int xa, int xb,
                                           float qx, float qy, float qz, Best& best);

// Phase A -- ball-bounded search, one query per lane.
//
// Two facts make the per-iteration search tiny:
//  (1) temporal coherence: the target is static during Align, so last iteration's match t_prev is
//      still a target point and |q - t_prev| bounds the new nearest distance from above;
//  (2) certified trimming: IcpFast only uses matches with d2 <= the 0.7-quantile
//      (icp_fast.cc:496-498).  A query whose nearest neighbour is provably farther than the radius
//      R_cap actually searched is recorded with the LOWER bound R_cap^2; as long as the quantile
//      comes out below every such bound (checked by nn_validate before anything consumes it) the
//      kept set, A, b, limit and score are exactly those of a full exact search.  If the check
//      fails (first iteration from a poor guess, dist_outlier_ratio = 1, ...) the ring search and
//      the brute-force fallback make every match exact, as they always do for find_closests.
// So each lane visits only the cells meeting the ball (q, min(|q - t_prev|, R_cap)): typically
// 1-2 cells per axis once ICP has settled.
//
//
//  (3) certificates: every search also records L = a lower bound on the distance from the query to every
//      target point OTHER than its match (runner-up distance, capped by the searched radius).  By a later
//      iteration the query has moved by at most delta = (pot_a now - pot_a then) |s| + (pot_b now - pot_b then)
//      (PairState::pot_a / pot_b: the running sums of ||dR||_F and |dt| of the pose updates; the record holds L plus
//      the potential at the time of the search, see with_pot / bound_now), so all other points are still at least
//      L - delta away: if the old match is closer than that it is provably still the unique nearest
//      neighbour and no search happens at all (nn_certify).  Lower-bounded ("hard") queries keep their
//      bound the same way.  Only the queries whose certificate fails are compacted into dlist and searched
//      (nn_ball_listed); iteration 0 searches everything.
// The search radius is min(R_cap, d_prev + margin) rather than d_prev so the runner-up information
// reaches beyond the match; margin = 3 x the query's motion in this iteration (1 cm .. 10 cm).
__device__ __forceinline__ float search_radius2(float r2cap, float dub2, float margin) {
  const float r = sqrtf(dub2) + margin;
  return fminf(r2cap, r * r);
}
// how far beyond the previous match the search looks: enough that next iteration's certificate
// (runner-up bound minus the query's motion) can still clear the match; motion shrinks every iteration
__device__ __forceinline__ float search_margin(float delta) { return fminf(fmaxf(3.0f * delta, 0.01f), 0.10f); }

__global__ __launch_bounds__(kNnThreads) void nn_ball(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int count = st->ns;
  const int base0 = blk * (kNnThreads * kBallItems);
  if (base0 >= count) return;
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t so = (size_t)pair * b.ns_cap;
  const uint2* __restrict__ words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* __restrict__ cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const float4* __restrict__ tq = b.tq + (size_t)pair * b.nt_cap;
  const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
  const float h = st->h, inv_h = st->inv_h;
  const int nx = st->nx, ny = st->ny, nz = st->nz, wx = st->wx;
  const float r2cap = st->rcap2;
  const bool have_prev = st->iter > 0;
  const Pot pot = {(float)st->pot_a, (float)st->pot_b, 0.f, 0.f};
  uint32_t min_lb = 0xffffffffu;

  for (int it = 0; it < kBallItems; ++it) {
    const int e = base0 + it * kNnThreads + threadIdx.x;
    bool hard = false;
    int i = -1;
    if (e < count) {
      i = e;
      double px, py, pz;
      const float4 s4 = ld_src(b, so + i);
      transform_point(st->M, s4, px, py, pz);
      const float qx = (float)px, qy = (float)py, qz = (float)pz;
      Best best = {INFINITY, -1, INFINITY};
      float d2out = INFINITY, lbout = 0.f;
      int jout = -1;
      if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
        float R2 = r2cap;
        int jp = -1;
        if (have_prev) {
          jp = b.idx[so + i];
          if (jp >= 0) {
            double ux, uy, uz;
            transform_point(st->M_prev, s4, ux, uy, uz);
            const float ex = qx - (float)ux, ey = qy - (float)uy, ez = qz - (float)uz;
            R2 = search_radius2(r2cap, dist2(tq[jp], qx, qy, qz), search_margin(sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)))));
          }
        }
        // every target point within sqrt(R2) of q lies in a cell meeting [q - Rs, q + Rs]^3
        const float Rs = sqrtf(R2) * 1.0001f + 1.0e-3f * h;
        const int x0 = max(cell_coord(qx - Rs, ox, inv_h), 0), x1 = min(cell_coord(qx + Rs, ox, inv_h), nx - 1);
        const int y0 = max(cell_coord(qy - Rs, oy, inv_h), 0), y1 = min(cell_coord(qy + Rs, oy, inv_h), ny - 1);
        const int z0 = max(cell_coord(qz - Rs, oz, inv_h), 0), z1 = min(cell_coord(qz + Rs, oz, inv_h), nz - 1);
        if (x0 <= x1) {
          const float slack = 2.0e-3f * h;
          for (int z = z0; z <= z1; ++z) {
            const float zl = oz + (float)z * h;
            const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + h)) - slack, 0.f);
            for (int y = y0; y <= y1; ++y) {
              const float yl = oy + (float)y * h;
              const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + h)) - slack, 0.f);
              if (fmaf(dy, dy, dz * dz) > R2) continue;            // the row lies outside the ball
              uint32_t sb, se;
              row_slots(words, (z * ny + y) * wx, x0, x1, sb, se);
              if (se > sb) {
                const uint32_t j0 = cstart[sb], j1 = cstart[se];
                for (uint32_t j = j0; j < j1; ++j) test_ascending_ru(tq[j], (int)j, qx, qy, qz, best);
              }
            }
          }
        }
        if (best.d2 <= R2) {                  // exact: everything within sqrt(R2) was seen
          d2out = best.d2;
          jout = best.j;
          lbout = sqrtf(fminf(best.s2, R2));  // every other point is at least this far
        } else {                              // certified lower bound: nothing lies within sqrt(R2)
          d2out = R2;
          jout = best.j >= 0 ? best.j : jp;   // an upper-bound seed for later iterations
          lbout = -sqrtf(R2);
          hard = true;
          min_lb = min(min_lb, __float_as_uint(R2));
        }
      }
      b.d2[so + i] = d2out;
      st_match(b, so + i, jout, with_pot(lbout, pot_at(pot, norm3(s4.x, s4.y, s4.z))));
      const uint32_t key = __float_as_uint(d2out);
      if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
    }
    const unsigned long long hm = __ballot(hard);
    if (hm) {
      uint32_t basepos = 0;
      if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
      basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
      if (hard) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = i;
    }
  }
  if (min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// The search of the queries nn_certify could not certify (dlist), on a small fixed grid per pair.  Once ICP has settled
// a pair has a few hundred to a few thousand of them and a query's search is a chain of dependent lookups per grid row
// (row words -> run bounds -> points), so the launch is pure latency: the grid's threads are therefore dealt out L at a
// time to one query (L = the largest power of two <= 16 the list leaves room for), the L lanes take the rows of the
// query's ball in turn and their results are merged with the tie rule of the sequential sweep (smallest distance, then
// smallest sorted position; runner-up = the smallest of the rest).  Same ids, distances and bounds as a one-lane sweep.
// (nn_ball_listed, the search of the queries whose certificate failed, follows `accumulate` below: its fused-path form sums too)

// SMHIP_NN_NABO: work classes of the queries to walk again (buckets scanned by their last walk: <= 2, 3-4, 5-7, more) and
// where the k-th member of class c sits: classes 0 / 1 fill dlist from its two ends, classes 2 / 3 hlist (a pair's
// lists hold ns_cap entries and the classes together at most ns, so the ends never meet)
__device__ __forceinline__ int nabo_class(uint32_t buckets) { return buckets <= 2u ? 0 : (buckets <= 4u ? 1 : (buckets <= 7u ? 2 : 3)); }
__device__ __forceinline__ int32_t* nabo_list_slot(const IcpDev& b, size_t so, int c, uint32_t k) {
  int32_t* base = (c < 2 ? b.dlist : b.hlist) + so;
  return (c & 1) ? base + (b.ns_cap - 1) - k : base + k;
}

// SMHIP_NN_NABO: the queries a wave's ITEMS rounds found to walk again (bit `it` of nabo_failed = this lane's query of round
// `it`) go to one of four lists by the number of buckets their last walk scanned: a wave of the list walk executes the union of
// its lanes' walks, so queries of like cost share waves (nabo_kernels.hip, nabo_class).  Done once per wave AFTER the streaming
// loop -- an atomic whose result is needed waits for every load issued before it, which would empty the two-deep pipeline in
// every round that has a failing certificate (nearly all of them in the first iterations): the classes of the failing queries
// (their loads all in flight together), four counts, four atomics, then the entries.
template <int ITEMS>
__device__ __forceinline__ void nabo_list_append(const IcpDev& b, PairState* st, size_t so, int base, int ns, uint32_t nabo_failed, int lane) {
  static_assert(ITEMS <= 32, "one bit per round");
  if (!__ballot(nabo_failed != 0u)) return;
  unsigned long long cls2 = 0;                                         // two class bits per round
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int i = base + it * kNnThreads + (int)threadIdx.x;
    const uint32_t w = (nabo_failed >> it) & 1u ? (uint32_t)b.nabo_work[so + min(i, ns - 1)] : 0u;
    cls2 |= (unsigned long long)nabo_class(w) << (2 * it);
  }
  uint32_t tot[4] = {0, 0, 0, 0};
  const unsigned long long lt = (1ull << lane) - 1ull;
  auto class_masks = [&](int it, unsigned long long* m) {
    const bool f = (nabo_failed >> it) & 1u;
    const uint32_t c = (uint32_t)(cls2 >> (2 * it)) & 3u;
    const unsigned long long fm = __ballot(f), b0 = __ballot(f && (c & 1u)), b1 = __ballot(f && (c & 2u));
    m[0] = fm & ~b0 & ~b1; m[1] = fm & b0 & ~b1; m[2] = fm & ~b0 & b1; m[3] = fm & b0 & b1;
    return fm;
  };
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    unsigned long long m[4];
    if (!class_masks(it, m)) continue;                                 // wave-uniform
#pragma unroll
    for (int c = 0; c < 4; ++c) tot[c] += (uint32_t)__popcll(m[c]);
  }
  uint32_t basepos[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uint32_t v = 0;
    if (lane == 0 && tot[c]) v = atomicAdd(&st->nabo_count[c], tot[c]);
    basepos[c] = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    unsigned long long m[4];
    if (!class_masks(it, m)) continue;
    const uint32_t c = (uint32_t)(cls2 >> (2 * it)) & 3u;
    if ((nabo_failed >> it) & 1u) {
      const unsigned long long mine = c == 0 ? m[0] : (c == 1 ? m[1] : (c == 2 ? m[2] : m[3]));
      const uint32_t bp = c == 0 ? basepos[0] : (c == 1 ? basepos[1] : (c == 2 ? basepos[2] : basepos[3]));
      *nabo_list_slot(b, so, (int)c, bp + (uint32_t)__popcll(mine & lt)) = base + it * kNnThreads + (int)threadIdx.x;
    }
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) basepos[c2] += (uint32_t)__popcll(m[c2]);
  }
}

// Certificate pass (iterations >= 1): no search, five memory operations per query.
// ITEMS = rounds of 256 queries per workgroup: kBallItems in batches, 1 where that would leave too few workgroups (one pair)
// NABO = true: the records are traversal certificates of the libnabo walk (nabo_kernels.hip): how far the query may move
// before any decision of its walk, or the winner among the entries it scanned, can change.  A query that has moved less
// keeps its id -- by the same walk, not because it is nearest -- and only its distance to that id is recomputed.
template <int ITEMS, bool NABO = false>
__global__ __launch_bounds__(kNnThreads) void nn_certify(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int ns = st->ns;
  const int base = blk * (kNnThreads * ITEMS);
  if (base >= ns) return;
  double Mc[12];                         // read before the first store: scalar loads, held in SGPRs (see nn_ball_lds)
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  const Pot pot = {(float)st->pot_a, (float)st->pot_b, 0.f, 0.f};
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t so = (size_t)pair * b.ns_cap;
  const float4* __restrict__ tq = b.tq + (size_t)pair * b.nt_cap;
  const float r_need = 0.9f * sqrtf(st->rcap2);     // a hard query's bound must stay well above the quantile
  uint32_t min_lb = 0xffffffffu;
  uint32_t nabo_failed = 0;                          // NABO: bit `it` = this lane's query of round `it` has to be walked again
  // two-deep software pipeline: a round's streamed values (point, bound, previous match id) are loaded two rounds ahead and
  // the gather of the previous match one round ahead, so no round waits for a load it has just issued
  int ic = min(base + (int)threadIdx.x, ns - 1);
  float4 s_1 = ld_src(b, so + ic);
  float l_1 = ld_lb(b, so + ic);
  int j_1 = b.idx[so + ic];
  ic = min(base + kNnThreads + (int)threadIdx.x, ns - 1);
  float4 s_2 = ld_src(b, so + ic);
  float l_2 = ld_lb(b, so + ic);
  int j_2 = b.idx[so + ic];
  float4 t_1 = tq[max(j_1, 0)];
  for (int it = 0; it < ITEMS; ++it) {
    const int i = base + it * kNnThreads + threadIdx.x;
    bool hard = false, fail = false;
    const float4 s = s_1;
    const float l = l_1;
    const int j = j_1;
    const float4 t = t_1;
    s_1 = s_2; l_1 = l_2; j_1 = j_2;
    if (it + 1 < ITEMS) t_1 = tq[max(j_1, 0)];
    if (it + 2 < ITEMS) {
      ic = min(i + 2 * kNnThreads, ns - 1);
      s_2 = ld_src(b, so + ic);
      l_2 = ld_lb(b, so + ic);
      j_2 = b.idx[so + ic];
    }
    if (i < ns) {
      double px, py, pz;
      transform_point(Mc, s, px, py, pz);
      const float qx = (float)px, qy = (float)py, qz = (float)pz;
      // all points other than the match (or all points, for a lower-bounded query) are still at least Lp away: the bound as
      // recorded, less what the query can have moved since (the pair's motion potential now; the one then is in the record)
      const float Lp = bound_now(l, pot_at(pot, norm3(s.x, s.y, s.z)));
      fail = true;
      if (NABO) {
        // what is left of the recorded slack: the record (slack + potential then) less the potential now, less the float
        // roundings of both sums (the same |s| and the same expression at both times, so nothing else differs), less two
        // roundings of q itself -- the walk compares floats derived from the ROUNDED position
        const float Pn = pot_at(pot, norm3(s.x, s.y, s.z));
        if (isfinite(qx) && isfinite(qy) && isfinite(qz) && l > 0.f && j >= 0 &&
            l - Pn - 4.0e-7f * (l + Pn) - 1.3e-7f * (fabsf(qx) + fabsf(qy) + fabsf(qz)) > 0.f) {
          const float d1 = dist2(t, qx, qy, qz);     // the bucket scan's arithmetic: the bits the walk would produce
          b.d2[so + i] = d1;
          atomicAdd(&s_hist[__float_as_uint(d1) >> kHistShift], 1u);
          fail = false;
        }
      } else if (isfinite(qx) && isfinite(qy) && isfinite(qz) && Lp > 0.f) {
        if (l > 0.f && j >= 0) {
          const float d1 = dist2(t, qx, qy, qz);
          if (d1 < Lp * Lp) {                       // still the unique nearest neighbour: exact, no search
            b.d2[so + i] = d1;
            atomicAdd(&s_hist[__float_as_uint(d1) >> kHistShift], 1u);
            fail = false;
          }
        } else if (l < 0.f && Lp >= r_need) {       // still provably farther than the trimming radius
          const float lb2 = Lp * Lp;
          b.d2[so + i] = lb2;
          atomicAdd(&s_hist[__float_as_uint(lb2) >> kHistShift], 1u);
          min_lb = min(min_lb, __float_as_uint(lb2));
          hard = true;
          fail = false;
        }
      }
    }
    const unsigned long long dm = __ballot(fail);
    if (NABO) {
      if (fail) nabo_failed |= 1u << it;               // (listed after the loop: nabo_list_append)
    } else if (dm) {
      uint32_t basepos = 0;
      if (lane == 0) basepos = atomicAdd(&st->deferred_count, (uint32_t)__popcll(dm));
      basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
      if (fail) b.dlist[so + basepos + __popcll(dm & ((1ull << lane) - 1ull))] = i;
    }
    const unsigned long long hm = __ballot(hard);
    if (hm) {
      uint32_t basepos = 0;
      if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
      basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
      if (hard) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = i;
    }
  }
  if (NABO) nabo_list_append<ITEMS>(b, st, so, base, ns, nabo_failed, lane);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) min_lb = min(min_lb, (uint32_t)__shfl_xor((int)min_lb, off, 64));
  if (lane == 0 && min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// nn_ball with the voxel lookups served from LDS.  The profile of nn_ball (profiles/r01b_pmc_*) shows
// the texture-address path ~85 % busy with ~38 divergent vector-memory instructions per 64 queries,
// a third of them the words -> cstart lookups.  Here a workgroup (256 Morton-consecutive queries)
// first takes the bounding box of all its search balls and stages, with coalesced loads, a dense table
//     tab[row][x] = position in the sorted target of the first point of the row's cells >= x
// for every grid row / column of that box (+1 column).  A lane's candidates for cells [xa, xb] of a row
// are then tab[row][xa] .. tab[row][xb + 1]: two LDS reads instead of 3-4 dependent global loads.
// Blocks whose box does not fit (sparse, far-range queries) use the global lookups of nn_ball.
//
// Latency shape (the kernel is latency-bound: waves spend ~80 % waiting): per round of 256 queries the
// dependent global-memory levels are  (1) src[i] and idx[i] together (prefetched one round ahead),
// (2) the previous match tq[idx] in flight WHILE the row tables are staged (words, then cstart) from
// a box that only needs q +- R_cap, (3) the candidates.  A workgroup runs kBallItems rounds so the
// histogram flush and the prefetch are amortised.
__device__ __forceinline__ void sweep_run(const float4* __restrict__ tq, uint32_t j0, uint32_t j1,
                                          float qx, float qy, float qz, Best& best) {
  uint32_t j = j0;
  for (; j + 4 <= j1; j += 4) {     // four independent loads in flight per lane
    const float4 t0 = tq[j], t1 = tq[j + 1], t2 = tq[j + 2], t3 = tq[j + 3];
    test_ascending_ru(t0, (int)j, qx, qy, qz, best);
    test_ascending_ru(t1, (int)j + 1, qx, qy, qz, best);
    test_ascending_ru(t2, (int)j + 2, qx, qy, qz, best);
    test_ascending_ru(t3, (int)j + 3, qx, qy, qz, best);
  }
  for (; j < j1; ++j) test_ascending_ru(tq[j], (int)j, qx, qy, qz, best);
}

// ITEMS = rounds of 256 queries per workgroup: 4 for batches (fewer histogram flushes, prefetch across rounds), 1 when a
// launch would otherwise have too few workgroups to fill the GPU (single pairs: 118 -> 469 workgroups for 120 k points).
// Diagnostic, compiled out by default (hipcc -DSMHIP_PHASE_TIMING=1, then run with SMHIP_DEBUG_FLAGS=16): per-wave s_memtime
// deltas per phase of a round, summed into 8 u64 counters that live in the (idle during the iterations) tpart scratch of
// slot 0; finalize prints their shares and clears them.  What it showed (64 pairs, iterations 1-4, nearly every query
// searching): prologue + prefetch 13 %, phase C + compaction 13 %, wait at barrier A 9 %, search-lane setup + box 12 %,
// staging between barriers B and C 22 %, the search itself 20 %, stores 5 %, histogram flush 6 % -- no phase dominates.
#ifndef SMHIP_PHASE_TIMING
#define SMHIP_PHASE_TIMING 0
#endif
#if SMHIP_PHASE_TIMING
#define SMHIP_PHASE(k) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tprev; tprev = now_; } } while (0)
#else
#define SMHIP_PHASE(k) do { } while (0)
#endif
// FIRST = the launch of iteration 0 (the host knows): no previous match, no certificate, one radius for every query -- the loads,
// the gather of the previous match and the certificate arithmetic drop out at compile time.
// The rounds of one workgroup: ITEMS rounds of 256 queries from source index base0 on.  `st` is the pair's state in global memory
// (the counters the rounds add to); `ls` is where the rounds READ the pair's pose-dependent state from -- the same object for the
// kernel below, the workgroup's own LDS copy for the single-pair persistent kernel (icp_one.hip), whose iterations never
// write the state back between rounds.  Mc = ls->M, already in registers.  Distances go to s_hist (the caller zeroes and flushes it).
template <int ITEMS, bool FIRST>
__device__ __forceinline__ void ball_lds_rounds(const IcpDev& b, PairState* st, const PairState* ls, int pair, int base0, const double* Mc,
                                                uint32_t* s_hist, uint32_t& min_lb) {
  const int ns = ls->ns;
  // the pair's motion potential (PairState::pot_a / pot_b) and the last iteration's step norms: how far a query can have moved
  // since its bound was recorded / in this iteration, from |s| alone -- no second transform
  const Pot pot = {(float)ls->pot_a, (float)ls->pot_b, (float)ls->step_a, (float)ls->step_b};
  const float Mtx = (float)Mc[3], Mty = (float)Mc[7], Mtz = (float)Mc[11];
#if SMHIP_PHASE_TIMING
  const bool timing = (b.debug_flags & 16) != 0;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = timing ? __builtin_readcyclecounter() : 0ull;
#endif
  __shared__ uint32_t s_tab[kLdsTableCap];
  __shared__ float4 s_pts[kLdsPointCap];
  __shared__ uint32_t s_roff[kLdsRowCap + 1];
  __shared__ uint32_t s_occ[kLdsRowCap / 32 + 1];   // bit r = staged row r of the box holds target points (+ one word so that two can always be read)
  __shared__ float4 s_q[kNnThreads];          // queries of the round that need a search: x, y, z, previous match
  __shared__ float s_r2[kNnThreads];          // their squared search radius
  __shared__ uint16_t s_lid[kNnThreads];      // their lane in the round
  __shared__ uint32_t s_nsearch[2];
  __shared__ uint32_t s_w[17];
  __shared__ int s_box[2][6];
  if (threadIdx.x < 6) { s_box[0][threadIdx.x] = threadIdx.x < 3 ? 0x3fffffff : -0x3fffffff; s_box[1][threadIdx.x] = s_box[0][threadIdx.x]; }
  if (threadIdx.x < 2) s_nsearch[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t so = (size_t)pair * b.ns_cap;
  const uint2* __restrict__ words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* __restrict__ cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const float4* __restrict__ tq = b.tq + (size_t)pair * b.nt_cap;
  const float ox = ls->origin[0], oy = ls->origin[1], oz = ls->origin[2];
  const float h = ls->h, inv_h = ls->inv_h;
  const int nx = ls->nx, ny = ls->ny, nz = ls->nz, wx = ls->wx, nw = ls->nw;
  const float r2cap = ls->rcap2;
  const float r_need = 0.9f * sqrtf(r2cap);
  const bool have_prev = !FIRST && ls->iter > 0;
  const bool certify = have_prev && b.certify;

  // level 1 of round 0
  int i = base0 + threadIdx.x;
  float4 s_cur = make_float4(0, 0, 0, 0);
  int jp_cur = -1;
  float l_cur = 0.f;
  if (i < ns) {
    s_cur = ld_src(b, so + i);
    if (have_prev) { jp_cur = b.idx[so + i]; if (certify) l_cur = ld_lb(b, so + i); }
  }

  for (int it = 0; it < ITEMS; ++it) {
    const int base = base0 + it * kNnThreads;
    if (base >= ns) break;                               // block-uniform
    i = base + threadIdx.x;
    int* box = s_box[it & 1];
    uint32_t* nsearch = &s_nsearch[it & 1];
    // prefetch level 1 of the next round
    const int i_next = i + kNnThreads;
    float4 s_next = make_float4(0, 0, 0, 0);
    int jp_next = -1;
    float l_next = 0.f;
    if (it + 1 < ITEMS && i_next < ns) {
      s_next = ld_src(b, so + i_next);
      if (have_prev) { jp_next = b.idx[so + i_next]; if (certify) l_next = ld_lb(b, so + i_next); }
    }
    SMHIP_PHASE(0);      // prologue / prefetch issue
    // ---- phase C (every lane): query, previous match, certificate
    float qx = 0.f, qy = 0.f, qz = 0.f;
    bool valid = false;
    if (i < ns) {
      double px, py, pz;
      transform_point(Mc, s_cur, px, py, pz);
      qx = (float)px; qy = (float)py; qz = (float)pz;
      valid = isfinite(qx) && isfinite(qy) && isfinite(qz);
    }
    const bool has_jp = valid && jp_cur >= 0;
    float dub2 = INFINITY;
    if (has_jp) dub2 = dist2(tq[jp_cur], qx, qy, qz);
    bool need_search = valid, hard = false;
    if (i < ns && !valid) {                                // NaN / inf input: no match
      b.d2[so + i] = INFINITY; st_match(b, so + i, -1, 0.f);
    }
    // (|s| and the other roots of this kernel by the hardware instruction, 1 ulp: each feeds a bound that bound_now / the cell
    // block take with 1e-5 .. 1e-4 of relative slack; the IEEE sequence is ~13 instructions a root in a kernel bound by their issue)
    const float sn = __builtin_amdgcn_sqrtf(fmaf(s_cur.z, s_cur.z, fmaf(s_cur.y, s_cur.y, s_cur.x * s_cur.x)));
    // an upper bound of this iteration's motion of the query (first iteration: no motion history yet)
    const float delta = have_prev ? fmaf(pot.sa, sn, pot.sb) : 0.03f;
    if (certify && valid) {
      // every point other than the match (every point, for a lower-bounded query) is still >= Lp away
      const float Lp = bound_now(l_cur, pot_at(pot, sn));
      if (Lp > 0.f) {
        if (l_cur > 0.f && has_jp && dub2 < Lp * Lp) {     // still the unique nearest neighbour: exact, no search
          b.d2[so + i] = dub2;
          atomicAdd(&s_hist[__float_as_uint(dub2) >> kHistShift], 1u);
          need_search = false;
        } else if (l_cur < 0.f && Lp >= r_need) {          // still provably beyond the trimming radius
          const float lb2 = Lp * Lp;
          b.d2[so + i] = lb2;
          atomicAdd(&s_hist[__float_as_uint(lb2) >> kHistShift], 1u);
          min_lb = min(min_lb, __float_as_uint(lb2));
          hard = true;
          need_search = false;
        }
      }
    }
    // compact the lanes that need a search into s_q (wave-aggregated, order-preserving inside a wave)
    {
      const unsigned long long sm = __ballot(need_search);
      if (sm) {
        uint32_t basepos = 0;
        if (lane == 0) basepos = atomicAdd(nsearch, (uint32_t)__popcll(sm));
        basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
        if (need_search) {
          const uint32_t pos = basepos + __popcll(sm & ((1ull << lane) - 1ull));
          s_q[pos] = make_float4(qx, qy, qz, __int_as_float(jp_cur));
          s_r2[pos] = has_jp ? search_radius2(r2cap, dub2, search_margin(delta)) : r2cap;
          s_lid[pos] = (uint16_t)threadIdx.x;
        }
      }
    }
    {   // lower-bounded lanes of phase C go to the hard list right away
      const unsigned long long hm = __ballot(hard);
      if (hm) {
        uint32_t basepos = 0;
        if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
        basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
        if (hard) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = i;
      }
    }
    SMHIP_PHASE(1);      // phase C + compaction
    __syncthreads();                                      // (A) search list complete
    SMHIP_PHASE(2);      // wait at barrier A
    const int nq = (int)*nsearch;
    if (threadIdx.x == 0 && nq) atomicAdd(&st->deferred_count, (uint32_t)nq);     // statistics only
    // ---- phase S: the first nq threads own one searching query each
    const bool mine = (int)threadIdx.x < nq;
    float R2 = 0.f;
    int jseed = -1, gi = -1;
    if (mine) {
      const float4 q4 = s_q[threadIdx.x];
      qx = q4.x; qy = q4.y; qz = q4.z; jseed = __float_as_int(q4.w);
      R2 = s_r2[threadIdx.x];
      gi = base + (int)s_lid[threadIdx.x];
    }
    int x0 = 1, x1 = 0, y0 = 1, y1 = 0, z0 = 1, z1 = 0;
    if (mine) {
      const float Rs = __builtin_amdgcn_sqrtf(R2) * 1.0001f + 1.0e-3f * h;
      x0 = max(cell_coord(qx - Rs, ox, inv_h), 0); x1 = min(cell_coord(qx + Rs, ox, inv_h), nx - 1);
      y0 = max(cell_coord(qy - Rs, oy, inv_h), 0); y1 = min(cell_coord(qy + Rs, oy, inv_h), ny - 1);
      z0 = max(cell_coord(qz - Rs, oz, inv_h), 0); z1 = min(cell_coord(qz + Rs, oz, inv_h), nz - 1);
    }
    const bool inbox = mine && x0 <= x1 && y0 <= y1 && z0 <= z1;
    if ((int)(threadIdx.x & ~63) < nq) {                  // waves that hold searching queries
      const int big = 0x3fffffff;
      const int mx0 = wave_min_i(inbox ? x0 : big), mx1 = wave_max_i(inbox ? x1 : -big);
      const int my0 = wave_min_i(inbox ? y0 : big), my1 = wave_max_i(inbox ? y1 : -big);
      const int mz0 = wave_min_i(inbox ? z0 : big), mz1 = wave_max_i(inbox ? z1 : -big);
      if (lane == 0 && mx0 <= mx1) {
        atomicMin(&box[0], mx0); atomicMin(&box[1], my0); atomicMin(&box[2], mz0);
        atomicMax(&box[3], mx1); atomicMax(&box[4], my1); atomicMax(&box[5], mz1);
      }
    }
    SMHIP_PHASE(3);      // search-lane setup + box
    __syncthreads();                                      // (B) box complete
    const int X0 = box[0], Y0 = box[1], Z0 = box[2];
    const int nxl = box[3] - X0 + 2, nyl = box[4] - Y0 + 1, nzl = box[5] - Z0 + 1;     // one extra x column
    const bool any = box[3] >= X0;
    const long long entries = any ? (long long)nxl * nyl * nzl : 0;
    const bool use_lds = any && entries <= kLdsTableCap;
    if (use_lds) {                                        // stage the row tables (all threads help)
      const float inv_nxl = 1.0f / (float)nxl, inv_nyl = 1.0f / (float)nyl;    // entries <= 2048: exact via float
      for (int e = threadIdx.x; e < (int)entries; e += kNnThreads) {
        const int r = (int)(((float)e + 0.5f) * inv_nxl);
        const int x = X0 + (e - r * nxl);
        const int zz = (int)(((float)r + 0.5f) * inv_nyl);
        const int y = Y0 + (r - zz * nyl), z = Z0 + zz;
        const int widx = (z * ny + y) * wx + (x >> 5);
        uint32_t val = (uint32_t)ls->nt;
        if (widx < nw) {
          const uint2 wd = words[widx];
          val = cstart[wd.y + __popc(wd.x & ((1u << (x & 31)) - 1u))];
        }
        s_tab[e] = val;
      }
    }
    // reset the other round's scratch while nobody reads it
    if (threadIdx.x < 6) s_box[(it + 1) & 1][threadIdx.x] = threadIdx.x < 3 ? 0x3fffffff : -0x3fffffff;
    if (threadIdx.x == 6) s_nsearch[(it + 1) & 1] = 0;
    // stage the box's target points: every grid row of the box is ONE contiguous run of tq
    const int rows = use_lds ? nyl * nzl : 0;
    bool pts_lds = use_lds && rows <= kLdsRowCap;
    if (use_lds) __syncthreads();                         // (B2) table visible
    if (pts_lds) {
      uint32_t len = 0;
      if ((int)threadIdx.x < rows) len = s_tab[threadIdx.x * nxl + nxl - 1] - s_tab[threadIdx.x * nxl];
      {   // which rows of the box hold a point at all: most (z, y) rows around a surface do not, and a query's row loop pays
          // ~50 vector instructions per row whether the row has candidates or not (kLdsRowCap = blockDim: thread r owns row r)
        const unsigned long long om = __ballot(len > 0);
        if (lane == 0) { s_occ[2 * (threadIdx.x >> 6)] = (uint32_t)om; s_occ[2 * (threadIdx.x >> 6) + 1] = (uint32_t)(om >> 32); }
        if (threadIdx.x == 0) s_occ[kLdsRowCap / 32] = 0u;
      }
      uint32_t total;
      const uint32_t off = block_excl_scan(len, s_w, &total);      // kLdsRowCap <= blockDim
      if ((int)threadIdx.x < rows) s_roff[threadIdx.x] = off;
      if (threadIdx.x == 0) s_roff[rows] = total;
      __syncthreads();
      pts_lds = total <= (uint32_t)kLdsPointCap;
      if (pts_lds) {
        for (uint32_t k = threadIdx.x; k < total; k += kNnThreads) {
          int lo = 0, hi = rows - 1;                               // last row with roff <= k
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_roff[mid] <= k) lo = mid; else hi = mid - 1; }
          const uint32_t j = s_tab[lo * nxl] + (k - s_roff[lo]);
          float4 t = tq[j];
          t.w = __int_as_float((int)j);
          s_pts[k] = t;
        }
      }
    }
    __syncthreads();                                      // (C) table / points staged
    SMHIP_PHASE(4);      // barrier B + staging + barrier C
    // search (threads < nq)
    bool hard2 = false;
    if (mine) {
      Best best = {INFINITY, -1, INFINITY};
      if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
        const float slack = 2.0e-3f * h;
        // The rows (z, y) of the query's block in ascending order.  With the box's points staged in LDS only the rows that hold
        // any are visited (s_occ): around a surface most rows of a ball are empty, and a row costs ~50 vector instructions before
        // its first candidate whether it has one or not.  (A block spans at most 31 rows in y: the search radius is capped at 14
        // cells, sync_options.)
        const uint32_t ymask = (2u << (y1 - y0)) - 1u;
        for (int z = z0; z <= z1; ++z) {
          const float zl = oz + (float)z * h;
          const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + h)) - slack, 0.f);
          const int ra = (z - Z0) * nyl + (y0 - Y0);                 // staged row of (z, y0) (LDS paths)
          uint32_t m = ymask;
          if (pts_lds) m &= __builtin_amdgcn_alignbit(s_occ[(ra >> 5) + 1], s_occ[ra >> 5], (uint32_t)(ra & 31));
          while (m) {
            const int bpos = __ffs((int)m) - 1;
            m &= m - 1u;
            const int y = y0 + bpos;
            const float yl = oy + (float)y * h;
            const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + h)) - slack, 0.f);
            if (fmaf(dy, dy, dz * dz) > R2) continue;              // the row lies outside the ball
            if (pts_lds) {
              const int r = ra + bpos;
              const int rb = r * nxl - X0;
              const uint32_t g0 = s_tab[rb + X0];
              const uint32_t k0 = s_roff[r] + (s_tab[rb + x0] - g0), k1 = s_roff[r] + (s_tab[rb + x1 + 1] - g0);
              for (const float4 *tp = s_pts + k0, *te = s_pts + k1; tp < te; ++tp) {     // one induction variable: the LDS address
                const float4 t = *tp;
                test_ascending_ru(t, __float_as_int(t.w), qx, qy, qz, best);
              }
            } else if (use_lds) {
              const int rb = (ra + bpos) * nxl - X0;
              sweep_run(tq, s_tab[rb + x0], s_tab[rb + x1 + 1], qx, qy, qz, best);
            } else {
              uint32_t sb, se;
              row_slots(words, (z * ny + y) * wx, x0, x1, sb, se);
              if (se > sb) sweep_run(tq, cstart[sb], cstart[se], qx, qy, qz, best);
            }
          }
        }
      }
      float d2out, lbout;
      int jout;
      if (best.d2 <= R2) {                  // exact: everything within sqrt(R2) was seen
        d2out = best.d2;
        jout = best.j;
        lbout = __builtin_amdgcn_sqrtf(fminf(best.s2, R2));  // every other point is at least this far
      } else {                              // certified lower bound
        d2out = R2;
        jout = best.j >= 0 ? best.j : jseed;
        lbout = -__builtin_amdgcn_sqrtf(R2);
        hard2 = true;
        min_lb = min(min_lb, __float_as_uint(R2));
      }
      SMHIP_PHASE(5);    // search
      b.d2[so + gi] = d2out;
      // |s| of the query from its transformed position: M is rigid, so |s| = |q - t|
      {
        const float ux = qx - Mtx, uy = qy - Mty, uz = qz - Mtz;
        st_match(b, so + gi, jout, with_pot(lbout, pot_at(pot, __builtin_amdgcn_sqrtf(fmaf(uz, uz, fmaf(uy, uy, ux * ux))))));
      }
      const uint32_t key = __float_as_uint(d2out);
      if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
    }
    {
      const unsigned long long hm = __ballot(hard2);
      if (hm) {
        uint32_t basepos = 0;
        if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
        basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
        if (hard2) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = gi;
      }
    }
    s_cur = s_next; jp_cur = jp_next; l_cur = l_next;
    SMHIP_PHASE(6);      // stores + hard list
  }
#if SMHIP_PHASE_TIMING
  if (timing && lane == 0) {
    unsigned long long* tc = reinterpret_cast<unsigned long long*>(b.tpart);
    for (int k = 0; k < 8; ++k) atomicAdd(&tc[k], tacc[k]);
  }
#endif
}

// the smallest lower bound a workgroup recorded -> PairState::min_lb_key; its histogram -> the pair's
__device__ __forceinline__ void flush_min_lb(PairState* st, uint32_t min_lb) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) min_lb = min(min_lb, (uint32_t)__shfl_xor((int)min_lb, off, 64));
  if ((threadIdx.x & 63) == 0 && min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
}
__device__ __forceinline__ void flush_hist(const IcpDev& b, int pair, const uint32_t* s_hist) {
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}
__device__ __forceinline__ void flush_min_lb_and_hist(const IcpDev& b, PairState* st, int pair, uint32_t min_lb, const uint32_t* s_hist) {
  flush_min_lb(st, min_lb);
  flush_hist(b, pair, s_hist);
}

template <int ITEMS, bool FIRST = false>
__global__ __launch_bounds__(kNnThreads, 5) void nn_ball_lds(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int base0 = blk * (kNnThreads * ITEMS);
  if (base0 >= st->ns) return;
  // The transform is read HERE, before the kernel's first store: the compiler then knows nothing has clobbered it,
  // loads it through the scalar cache into SGPRs once, and the rounds below transform from registers.  Read inside the
  // loop (behind the rounds' global stores) each lane fetched the same 96 bytes with six 16-byte vector loads per round.
  double Mc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;   // (visible to the rounds after their first barrier)
  uint32_t min_lb = 0xffffffffu;
  ball_lds_rounds<ITEMS, FIRST>(b, st, st, pair, base0, Mc, s_hist, min_lb);
  // statistics: queries searched by this block are counted through deferred_count
  flush_min_lb_and_hist(b, st, pair, min_lb, s_hist);
}

// ------------------------------------------------------------------------------------------
// nn_ball_wave: the search of the iterations in which (nearly) every query searches, one WAVE per 64 Morton-consecutive
// queries, every candidate broadcast to the whole wave.
//
// nn_ball_lds gives every query its own walk over the rows of its ball: ~50 vector instructions of bookkeeping per row
// visited and a candidate loop whose trip count is the wave's longest run, per row -- a wave pays the sum over row steps of
// the maxima.  Counted on the bench scans (64 consecutive queries, 0.3 m balls, 0.25 m cells): a lane has 35 candidates in 6
// occupied rows of its own, the wave executes ~54 candidate steps + 7.5 row steps + staging and compaction, ~1 900 vector
// instructions per 64 queries in all (SQ_INSTS_VALU).  The bounding box of the wave's balls holds 110 target points (median
// 89) in 11 occupied rows of 49.  So here the WAVE walks that box once: lane r looks up the run of row r (two dependent loads
// for all rows at a time instead of per row and query), the runs' points are copied -- coalesced, every lane a point -- into
// the wave's own LDS strip as PAIRS {x0 x1 y0 y1 | z0 z1 j0 j1}, and all 64 lanes test pair after pair from two broadcast
// reads with packed fp32 arithmetic: 6 v_pk instructions for two squared distances + 4 per candidate for (nearest, runner-
// up, position) = 7 vector instructions per candidate, no per-query bookkeeping, no divergence, no workgroup barrier -- the
// four waves of a workgroup only share the histogram.  (First built with the candidates fetched by SCALAR loads into SGPR
// operands -- no LDS at all: 2 x slower than nn_ball_lds; the scalar cache serves a few loads at a time and every wave sat
// waiting for it.)
//
// Same results as the per-query walks: candidates are met in ascending sorted position (rows in (z, y) order, a row's run in
// x order), so the tie rule holds; a point outside a query's own cube is farther than its radius, so it neither becomes the
// match of an exact query nor lowers min(runner-up, radius).  Only the SEED a lower-bounded query keeps (the nearest point
// seen, all of them beyond the radius) can be another point -- it only sizes a later search.
__device__ __forceinline__ void listed_append_hard(const IcpDev& b, PairState* st, size_t so, bool hard, int i, int lane);
constexpr int kWaveBoxRows = 1024;        // grid rows of a wave's box it walks cooperatively; beyond: every lane walks its own ball
constexpr int kWaveCand = 256;            // candidates a wave stages per pass (4 KiB of LDS per wave)

// one query's own walk, lookups from global memory (nn_ball's loop): the fallback for a wave whose queries are far apart
__device__ __forceinline__ void lane_ball_search(const uint2* __restrict__ words, const uint32_t* __restrict__ cstart, const float4* __restrict__ tq,
                                                 float ox, float oy, float oz, float h, int ny, int wx, float qx, float qy, float qz, float R2,
                                                 int x0, int x1, int y0, int y1, int z0, int z1, Best& best) {
  const float slack = 2.0e-3f * h;
  for (int z = z0; z <= z1; ++z) {
    const float zl = oz + (float)z * h;
    const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + h)) - slack, 0.f);
    for (int y = y0; y <= y1; ++y) {
      const float yl = oy + (float)y * h;
      const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + h)) - slack, 0.f);
      if (fmaf(dy, dy, dz * dz) > R2) continue;            // the row lies outside the ball
      uint32_t sb, se;
      row_slots(words, (z * ny + y) * wx, x0, x1, sb, se);
      if (se > sb) {
        const uint32_t j0 = cstart[sb], j1 = cstart[se];
        for (uint32_t j = j0; j < j1; ++j) test_ascending_ru(tq[j], (int)j, qx, qy, qz, best);
      }
    }
  }
}
// a staged pair against the lane's query: dist2's expression on both halves at once (v_pk_add / mul / fma_f32), then the
// sweep's update for the first and the second (ascending positions: strict "<")
__device__ __forceinline__ void test_pair_ru(const float4 a, const float4 c, float qx, float qy, float qz, Best& best) {
  const f32x2 dx = (f32x2){qx, qx} - (f32x2){a.x, a.y};
  const f32x2 dy = (f32x2){qy, qy} - (f32x2){a.z, a.w};
  const f32x2 dz = (f32x2){qz, qz} - (f32x2){c.x, c.y};
  f32x2 d = dx * dx;
  d = __builtin_elementwise_fma(dy, dy, d);
  d = __builtin_elementwise_fma(dz, dz, d);
  best.s2 = __builtin_amdgcn_fmed3f(d.x, best.d2, best.s2);
  if (d.x < best.d2) { best.d2 = d.x; best.j = __float_as_int(c.z); }
  best.s2 = __builtin_amdgcn_fmed3f(d.y, best.d2, best.s2);
  if (d.y < best.d2) { best.d2 = d.y; best.j = __float_as_int(c.w); }
}

template <int ITEMS, bool FIRST>
__global__ __launch_bounds__(kNnThreads) void nn_ball_wave(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int ns = st->ns;
  const int base0 = blk * (kNnThreads * ITEMS);
  if (base0 >= ns) return;
  double Mc[12];                         // read before the first store: scalar loads, held in SGPRs (see nn_ball_lds)
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  const Pot pot = {(float)st->pot_a, (float)st->pot_b, (float)st->step_a, (float)st->step_b};
  const float Mtx = (float)Mc[3], Mty = (float)Mc[7], Mtz = (float)Mc[11];
  __shared__ uint32_t s_hist[kHistBins];
  __shared__ float4 s_cand[kNnThreads / 64][kWaveCand];          // per wave: kWaveCand / 2 pairs of two float4
  __shared__ uint32_t s_roff[kNnThreads / 64][64];               // per wave: candidates before row r of the current 64 rows
  __shared__ uint32_t s_rbase[kNnThreads / 64][64];              //           sorted position of the row's first point, less that
  __shared__ uint32_t s_nsearch;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  if (threadIdx.x == 0) s_nsearch = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* cand = s_cand[wave];
  uint32_t* roff = s_roff[wave];
  uint32_t* rbase = s_rbase[wave];
  const size_t so = (size_t)pair * b.ns_cap;
  const uint2* __restrict__ words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* __restrict__ cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const float4* __restrict__ tq = b.tq + (size_t)pair * b.nt_cap;
  const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
  const float h = st->h, inv_h = st->inv_h;
  const int nx = st->nx, ny = st->ny, nz = st->nz, wx = st->wx;
  const float r2cap = st->rcap2;
  const float r_need = 0.9f * sqrtf(r2cap);
  const bool have_prev = !FIRST && st->iter > 0;
  const bool certify = have_prev && b.certify;
  uint32_t min_lb = 0xffffffffu;
  uint32_t nsearch = 0;

  int i = base0 + threadIdx.x;
  float4 s_cur = make_float4(0, 0, 0, 0);
  int jp_cur = -1;
  float l_cur = 0.f;
  if (i < ns) {
    s_cur = ld_src(b, so + i);
    if (have_prev) { jp_cur = b.idx[so + i]; if (certify) l_cur = ld_lb(b, so + i); }
  }
  for (int it = 0; it < ITEMS; ++it) {
    const int base = base0 + it * kNnThreads;
    if (base >= ns) break;                               // block-uniform
    i = base + threadIdx.x;
    const int i_next = i + kNnThreads;                   // the next round's streamed values, a round ahead
    float4 s_next = make_float4(0, 0, 0, 0);
    int jp_next = -1;
    float l_next = 0.f;
    if (it + 1 < ITEMS && i_next < ns) {
      s_next = ld_src(b, so + i_next);
      if (have_prev) { jp_next = b.idx[so + i_next]; if (certify) l_next = ld_lb(b, so + i_next); }
    }
    // ---- every lane: query, previous match, certificate (as nn_ball_lds's phase C)
    float qx = 0.f, qy = 0.f, qz = 0.f;
    bool valid = false;
    if (i < ns) {
      double px, py, pz;
      transform_point(Mc, s_cur, px, py, pz);
      qx = (float)px; qy = (float)py; qz = (float)pz;
      valid = isfinite(qx) && isfinite(qy) && isfinite(qz);
    }
    const bool has_jp = valid && jp_cur >= 0;
    float dub2 = INFINITY;
    if (has_jp) dub2 = dist2(tq[jp_cur], qx, qy, qz);
    bool need_search = valid, hard = false;
    if (i < ns && !valid) {                                // NaN / inf input: no match
      b.d2[so + i] = INFINITY; st_match(b, so + i, -1, 0.f);
    }
    float R2 = r2cap;
    if (!FIRST) {
      const float sn = __builtin_amdgcn_sqrtf(fmaf(s_cur.z, s_cur.z, fmaf(s_cur.y, s_cur.y, s_cur.x * s_cur.x)));
      const float delta = have_prev ? fmaf(pot.sa, sn, pot.sb) : 0.03f;
      if (certify && valid) {
        const float Lp = bound_now(l_cur, pot_at(pot, sn));
        if (Lp > 0.f) {
          if (l_cur > 0.f && has_jp && dub2 < Lp * Lp) {     // still the unique nearest neighbour: exact, no search
            b.d2[so + i] = dub2;
            atomicAdd(&s_hist[__float_as_uint(dub2) >> kHistShift], 1u);
            need_search = false;
          } else if (l_cur < 0.f && Lp >= r_need) {          // still provably beyond the trimming radius
            const float lb2 = Lp * Lp;
            b.d2[so + i] = lb2;
            atomicAdd(&s_hist[__float_as_uint(lb2) >> kHistShift], 1u);
            min_lb = min(min_lb, __float_as_uint(lb2));
            hard = true;
            need_search = false;
          }
        }
      }
      if (has_jp) R2 = search_radius2(r2cap, dub2, search_margin(delta));
      listed_append_hard(b, st, so, hard, i, lane);          // lower-bounded lanes keep their place in the hard list
    }
    const unsigned long long sm = __ballot(need_search);
    if (sm) {                                              // wave-uniform
      nsearch += (uint32_t)__popcll(sm);
      int x0 = 1, x1 = 0, y0 = 1, y1 = 0, z0 = 1, z1 = 0;
      if (need_search) {
        const float Rs = __builtin_amdgcn_sqrtf(R2) * 1.0001f + 1.0e-3f * h;
        x0 = max(cell_coord(qx - Rs, ox, inv_h), 0); x1 = min(cell_coord(qx + Rs, ox, inv_h), nx - 1);
        y0 = max(cell_coord(qy - Rs, oy, inv_h), 0); y1 = min(cell_coord(qy + Rs, oy, inv_h), ny - 1);
        z0 = max(cell_coord(qz - Rs, oz, inv_h), 0); z1 = min(cell_coord(qz + Rs, oz, inv_h), nz - 1);
      }
      const bool inbox = need_search && x0 <= x1 && y0 <= y1 && z0 <= z1;
      const int big = 0x3fffffff;
      const int X0 = wave_min_i(inbox ? x0 : big), X1 = wave_max_i(inbox ? x1 : -big);
      const int Y0 = wave_min_i(inbox ? y0 : big), Y1 = wave_max_i(inbox ? y1 : -big);
      const int Z0 = wave_min_i(inbox ? z0 : big), Z1 = wave_max_i(inbox ? z1 : -big);
      Best best = {INFINITY, -1, INFINITY};
      if (X0 <= X1) {                                      // some ball meets the grid
        const int nyl = Y1 - Y0 + 1;
        const long long nrows_ll = (long long)nyl * (Z1 - Z0 + 1);
        if (nrows_ll <= kWaveBoxRows) {
          const int nrows = (int)nrows_ll;
          const float inv_nyl = 1.0f / (float)nyl;
          for (int r0 = 0; r0 < nrows; r0 += 64) {         // 64 rows of the box at a time, one per lane
            const int r = r0 + lane;
            uint32_t j0 = 0, len = 0;
            if (r < nrows) {
              const int zz = (int)(((float)r + 0.5f) * inv_nyl);     // r / nyl (r < 1024: exact via float)
              const int y = Y0 + (r - zz * nyl), z = Z0 + zz;
              uint32_t sb, se;
              row_slots(words, (z * ny + y) * wx, X0, X1, sb, se);
              if (se > sb) { j0 = cstart[sb]; len = cstart[se] - j0; }
            }
            const uint32_t incl = wave_incl_scan(len, lane);
            const int N = __builtin_amdgcn_readlane((int)incl, 63);
            if (N == 0) continue;                          // wave-uniform
            roff[lane] = incl - len;
            rbase[lane] = j0 - (incl - len);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int p0 = 0; p0 < N; p0 += kWaveCand) {    // the rows' points in ascending sorted position, kWaveCand per pass
              const int cnt = min(kWaveCand, N - p0);
              // copy: flat candidate k of the 64 rows sits in the last row whose offset is <= k (empty rows share their successor's)
#pragma unroll
              for (int m = 0; m < kWaveCand / 64; ++m) {
                if (64 * m >= cnt) break;                  // wave-uniform
                const int c = lane + 64 * m;
                const uint32_t k = (uint32_t)(p0 + min(c, cnt - 1));
                int rho = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) rho = roff[min(rho + step, 63)] <= k && rho + step < 64 ? rho + step : rho;
                const uint32_t j = rbase[rho] + k;
                const float4 t = tq[j];
                if (c < cnt) {
                  float* o = reinterpret_cast<float*>(cand + 2 * (c >> 1)) + (c & 1);
                  o[0] = t.x; o[2] = t.y; o[4] = t.z; o[6] = __int_as_float((int)j);
                }
              }
              if ((cnt & 1) && lane == 0) {                // an odd pass: the last pair's second half is a point at infinity
                float* o = reinterpret_cast<float*>(cand + 2 * (cnt >> 1)) + 1;
                o[0] = 3.0e38f; o[2] = 0.f; o[4] = 0.f; o[6] = __int_as_float(-1);
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              const int npair = (cnt + 1) >> 1;
#pragma unroll 4
              for (int pp = 0; pp < npair; ++pp) test_pair_ru(cand[2 * pp], cand[2 * pp + 1], qx, qy, qz, best);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
        } else if (inbox) {
          lane_ball_search(words, cstart, tq, ox, oy, oz, h, ny, wx, qx, qy, qz, R2, x0, x1, y0, y1, z0, z1, best);
        }
      }
      bool hard2 = false;
      if (need_search) {
        float d2out, lbout;
        int jout;
        if (best.d2 <= R2) {                  // exact: everything within sqrt(R2) was seen
          d2out = best.d2;
          jout = best.j;
          lbout = __builtin_amdgcn_sqrtf(fminf(best.s2, R2));  // every other point is at least this far
        } else {                              // certified lower bound
          d2out = R2;
          jout = best.j >= 0 ? best.j : jp_cur;
          lbout = -__builtin_amdgcn_sqrtf(R2);
          hard2 = true;
          min_lb = min(min_lb, __float_as_uint(R2));
        }
        b.d2[so + i] = d2out;
        {   // |s| as nn_ball_lds's search lanes take it (from the moved point: M is rigid), so that the record has the same bits
          const float ux = qx - Mtx, uy = qy - Mty, uz = qz - Mtz;
          st_match(b, so + i, jout, with_pot(lbout, pot_at(pot, __builtin_amdgcn_sqrtf(fmaf(uz, uz, fmaf(uy, uy, ux * ux))))));
        }
        const uint32_t key = __float_as_uint(d2out);
        if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
      }
      listed_append_hard(b, st, so, hard2, i, lane);
    }
    s_cur = s_next; jp_cur = jp_next; l_cur = l_next;
  }
  if (lane == 0 && nsearch) atomicAdd(&s_nsearch, nsearch);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) min_lb = min(min_lb, (uint32_t)__shfl_xor((int)min_lb, off, 64));
  if (lane == 0 && min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
  __syncthreads();
  if (threadIdx.x == 0 && s_nsearch) atomicAdd(&st->deferred_count, s_nsearch);      // statistics only
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// a run of the sorted target tested with the explicit tie rule, four independent loads in flight per lane (one load per
// trip left every candidate a full memory latency: 400 us for the first ring of 120 k queries against a dense submap)
__device__ __forceinline__ void sweep_run_any(const float4* __restrict__ tq, uint32_t j0, uint32_t j1,
                                              float qx, float qy, float qz, Best& best) {
  uint32_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    const float4 t0 = tq[j], t1 = tq[j + 1], t2 = tq[j + 2], t3 = tq[j + 3];
    test_any_order(t0, (int)j, qx, qy, qz, best);
    test_any_order(t1, (int)j + 1, qx, qy, qz, best);
    test_any_order(t2, (int)j + 2, qx, qy, qz, best);
    test_any_order(t3, (int)j + 3, qx, qy, qz, best);
  }
  for (; j < j1; ++j) test_any_order(tq[j], (int)j, qx, qy, qz, best);
}

// candidates of the cells x in [xa, xb] of row (y, z): one contiguous run of the sorted target
__device__ __forceinline__ void search_row(const uint2* __restrict__ words, const uint32_t* __restrict__ cstart,
                                           const float4* __restrict__ tq, int rowbase, int xa, int xb,
                                           float qx, float qy, float qz, Best& best) {
  uint32_t sb, se;
  row_slots(words, rowbase, xa, xb, sb, se);
  if (se > sb) {
    sweep_run_any(tq, cstart[sb], cstart[se], qx, qy, qz, best);
  }
}

// The caller only uses matches closer than sqrt(nn_cutoff2) (GICP's correspondence distance): once the searched block
// guarantees that everything outside it is farther than that and nothing inside it is closer either, the query is done
// -- whatever its true nearest neighbour is, it is beyond the cutoff (the reported d2 is >= the cutoff too).
__device__ __forceinline__ bool ring_irrelevant(const IcpDev& b, float g, float best_d2) {
  return b.nn_cutoff2 > 0.f && g * g >= b.nn_cutoff2 && best_d2 >= b.nn_cutoff2;
}

// One query's exact ring search: rings r = 1, 2, 4, ... <= max_ring of cells around the query's cell, cells covered by an earlier
// ring skipped.  Returns false when the rings end without a guarantee (the brute-force fallback's case).  `st`: where the grid
// geometry is read from (the pair's state, or a workgroup's LDS copy of it).
__device__ __forceinline__ bool ring_search_query(const IcpDev& b, const PairState* st, int pair, float qx, float qy, float qz, Best& best) {
  bool resolved = true;
  if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
    const uint2* words = b.words + (size_t)pair * kMaxGridWords;
    const uint32_t* cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
    const float4* tq = b.tq + (size_t)pair * b.nt_cap;
    const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
    const float inv_h = st->inv_h;
    const int nx = st->nx, ny = st->ny, nz = st->nz, wx = st->wx;
    const int cx = cell_coord(qx, ox, inv_h), cy = cell_coord(qy, oy, inv_h), cz = cell_coord(qz, oz, inv_h);
    int rp = 0;           // radius already covered
    resolved = false;
    for (int r = 1; r <= b.max_ring; r *= 2) {
      const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
      const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
      const int z0 = max(cz - r, 0), z1 = min(cz + r, nz - 1);
      if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
        for (int z = z0; z <= z1; ++z)
          for (int y = y0; y <= y1; ++y) {
            const int rowbase = (z * ny + y) * wx;
            const bool inner = rp > 0 && abs(y - cy) <= rp && abs(z - cz) <= rp;
            if (!inner) {
              search_row(words, cstart, tq, rowbase, x0, x1, qx, qy, qz, best);
            } else {
              const int xl1 = min(cx - rp - 1, nx - 1), xr0 = max(cx + rp + 1, 0);
              if (x0 <= xl1) search_row(words, cstart, tq, rowbase, x0, xl1, qx, qy, qz, best);
              if (xr0 <= x1) search_row(words, cstart, tq, rowbase, xr0, x1, qx, qy, qz, best);
            }
          }
      }
      const float g = block_guarantee(st, qx, qy, qz, cx - r, cx + r, cy - r, cy + r, cz - r, cz + r);
      if (g == INFINITY || (g > 0.f && (best.d2 <= g * g || ring_irrelevant(b, g, best.d2)))) { resolved = true; break; }
      rp = r;
    }
  }
  return resolved;
}

// Phase B -- exact per-query ring search.  HARD = true: refinement of the lower-bounded queries of
// nn_ball, run only when nn_validate found that the quantile may reach one of the bounds (or when
// every match must be exact); HARD = false: over every source point (nn_ball skipped).
// Rings r = 1, 2, 4, ... <= max_ring; cells covered by an earlier ring are skipped.  What is still
// uncertified afterwards goes to the brute-force fallback list.
// blk / nblk: this workgroup's place among the workgroups that share the pair's list (the kernel below: blockIdx.x / gridDim.x;
// nn_refine_one: 0 / 1)
template <bool HARD>
__device__ __forceinline__ void ring_body(const IcpDev& b, PairState* st, int pair, int blk, int nblk, uint32_t* s_hist) {
  const int count = HARD ? (int)st->hard_count : st->ns;
  if (blk * kNnThreads >= count) return;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int e = blk * kNnThreads + threadIdx.x; e < count; e += nblk * kNnThreads) {
    const size_t so = (size_t)pair * b.ns_cap;
    const int i = HARD ? b.hlist[so + e] : e;
    if (HARD) {   // take the lower bound back out of the histogram
      const uint32_t old = __float_as_uint(b.d2[so + i]);
      if (old < 0x7f800000u) atomicSub(&gh[old >> kHistShift], 1u);
    }
    double px, py, pz;
    transform_point(st->M, ld_src(b, so + i), px, py, pz);
    const float qx = (float)px, qy = (float)py, qz = (float)pz;
    Best best = {INFINITY, -1, INFINITY};
    const bool resolved = ring_search_query(b, st, pair, qx, qy, qz, best);
    b.d2[so + i] = best.d2;
    st_match(b, so + i, best.j, 0.f);      // exact match, but no runner-up information: searched again next time
    if (resolved) {
      const uint32_t key = __float_as_uint(best.d2);
      if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
    } else {
      const uint32_t pos = atomicAdd(&st->unresolved_count, 1u);
      b.ulist[so + pos] = i;
      b.ukeys[so + pos] = ~0ull;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}
template <bool HARD>
__global__ __launch_bounds__(kNnThreads) void nn_ring(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  if (st->done) return;
  if (HARD && !st->refine) return;
  __shared__ uint32_t s_hist[kHistBins];
  ring_body<HARD>(b, st, pair, (int)blockIdx.x, (int)gridDim.x, s_hist);
}

// The same exact ring search with kCoopLanes adjacent lanes per query (rows of the cell block dealt round-robin, the
// group's bests merged with the usual tie rule): a single cloud of ~100 k queries gives the one-query-per-lane form
// fewer than two waves per SIMD, which cannot hide the latency of its dependent loads; this form has four times the
// waves.  Used for launches with few queries (the NDT / GICP fitness and correspondence passes, single pairs).
constexpr int kCoopLanes = 4;
__global__ __launch_bounds__(kNnThreads) void nn_ring_coop(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int count = st->ns;
  constexpr int kQueriesPerBlock = kNnThreads / kCoopLanes;
  if ((int)(blockIdx.x * kQueriesPerBlock) >= count) return;
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  const int sub = threadIdx.x & (kCoopLanes - 1);
  const size_t so = (size_t)pair * b.ns_cap;
  const uint2* words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const float4* tq = b.tq + (size_t)pair * b.nt_cap;
  const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
  const float inv_h = st->inv_h;
  const int nx = st->nx, ny = st->ny, nz = st->nz, wx = st->wx;
  // the loop bound is rounded up so that all lanes of a group (and a wave) take the same number of trips
  const int trips = (count + gridDim.x * kQueriesPerBlock - 1) / (gridDim.x * kQueriesPerBlock);
  for (int trip = 0; trip < trips; ++trip) {
    const int e = (trip * gridDim.x + blockIdx.x) * kQueriesPerBlock + (threadIdx.x / kCoopLanes);
    const bool live = e < count;
    const int i = live ? e : 0;
    double px, py, pz;
    transform_point(st->M, ld_src(b, so + i), px, py, pz);
    const float qx = (float)px, qy = (float)py, qz = (float)pz;
    Best best = {INFINITY, -1, INFINITY};
    bool resolved = true;
    if (live && isfinite(qx) && isfinite(qy) && isfinite(qz)) {
      // first ring only (where nearly every query ends; the few that need a wider block go to nn_ring_wide, a wave each:
      // left here they took 20-70 rows per lane and every workgroup waited for its slowest query)
      const int cx = cell_coord(qx, ox, inv_h), cy = cell_coord(qy, oy, inv_h), cz = cell_coord(qz, oz, inv_h);
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, nx - 1);
      const int y0 = max(cy - 1, 0), y1 = min(cy + 1, ny - 1);
      const int z0 = max(cz - 1, 0), z1 = min(cz + 1, nz - 1);
      if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
        // The query's OWN cell first, its points dealt out to the group's lanes four apart: against a dense submap (hundreds of
        // points per cell near the sensor) it nearly always holds the neighbour, and with that distance in hand every other
        // piece of the 3 x 3 x 3 block -- the two other cells of the own row, the eight other rows -- is cut down to the cells
        // the ball of that distance reaches before anything of it is loaded: what lies strictly farther than the best so far
        // cannot improve on it, nor tie with it.  (Rows used to be dealt out blind: a lane swept up to three full rows of three
        // cells whatever the first one found.)
        const float hh = st->h, slack = 2.0e-3f * hh;
        const bool own = cx >= 0 && cx < nx && cy >= 0 && cy < ny && cz >= 0 && cz < nz;   // (a query outside the grid has no cell of its own)
        if (own) {
          uint32_t sb, se;
          row_slots(words, (cz * ny + cy) * wx, cx, cx, sb, se);
          if (se > sb) {
            const uint32_t j0 = cstart[sb], j1 = cstart[se];
            for (uint32_t j = j0 + (uint32_t)sub; j < j1; j += kCoopLanes) test_any_order(tq[j], (int)j, qx, qy, qz, best);
          }
        }
#pragma unroll
        for (int off = 1; off < kCoopLanes; off <<= 1) {
          const float od = __shfl_xor(best.d2, off, 64);
          const int oj = __shfl_xor(best.j, off, 64);
          if (od < best.d2 || (od == best.d2 && oj >= 0 && (best.j < 0 || oj < best.j))) { best.d2 = od; best.j = oj; }
        }
        // the pieces left: for every row of the block its cells inside the ball, the own row in two parts around the own cell;
        // at most 10, round-robin over the lanes, <= 3 per lane, their lookups issued together
        uint32_t ja[3] = {0, 0, 0}, jb[3] = {0, 0, 0};
        uint2 wa[3], wc[3];
        int xa[3] = {0, 0, 0}, xb[3] = {0, 0, 0};
        int nrow = 0;
        {
          int k = 0;
          for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
              const bool own_row = own && z == cz && y == cy;
              const float zl = oz + (float)z * hh, yl = oy + (float)y * hh;
              const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + hh)) - slack, 0.f);
              const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + hh)) - slack, 0.f);
              const float rest = best.d2 - fmaf(dy, dy, dz * dz);
              int ca = x0, cb = x1;
              if (rest < 1.0e30f && rest >= 0.f) {
                const float xr = sqrtf(rest) * 1.0001f + slack;
                ca = max(x0, cell_coord(qx - xr, ox, inv_h)); cb = min(x1, cell_coord(qx + xr, ox, inv_h));
              }
              const int rowbase = (z * ny + y) * wx;
#pragma unroll
              for (int part = 0; part < 2; ++part) {                   // a row other than the own one is one part
                if (part == 1 && !own_row) break;
                int pa = ca, pb = cb;
                if (own_row) { if (part == 0) pb = min(cb, cx - 1); else pa = max(ca, cx + 1); }
                const bool mine = (k & (kCoopLanes - 1)) == sub;
                ++k;
                if (!mine || nrow >= 3 || rest < 0.f || pa > pb) continue;
                wa[nrow] = words[rowbase + (pa >> 5)];
                wc[nrow] = words[rowbase + (pb >> 5)];
                xa[nrow] = pa; xb[nrow] = pb;
                ++nrow;
              }
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (m < nrow) {
            const uint32_t sb = wa[m].y + __popc(wa[m].x & ((1u << (xa[m] & 31)) - 1u));
            const uint32_t se = wc[m].y + __popc(wc[m].x & (0xffffffffu >> (31 - (xb[m] & 31))));
            ja[m] = cstart[sb]; jb[m] = cstart[se];
          }
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (m < nrow)
            sweep_run_any(tq, ja[m], jb[m], qx, qy, qz, best);
      }
      // merge the group's bests: smaller distance, then smaller position
#pragma unroll
      for (int off = 1; off < kCoopLanes; off <<= 1) {
        const float od = __shfl_xor(best.d2, off, 64);
        const int oj = __shfl_xor(best.j, off, 64);
        if (od < best.d2 || (od == best.d2 && oj >= 0 && (best.j < 0 || oj < best.j))) { best.d2 = od; best.j = oj; }
      }
      const float g = block_guarantee(st, qx, qy, qz, cx - 1, cx + 1, cy - 1, cy + 1, cz - 1, cz + 1);
      resolved = g == INFINITY || (g > 0.f && best.d2 <= g * g);
    }
    if (live && sub == 0) {
      b.d2[so + i] = best.d2;
      st_match(b, so + i, best.j, 0.f);
      if (resolved) {
        const uint32_t key = __float_as_uint(best.d2);
        if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
      } else if (b.max_ring >= 2) {                       // wider rings: nn_ring_wide
        b.dlist[so + atomicAdd(&st->deferred_count, 1u)] = i;
      } else {
        const uint32_t pos = atomicAdd(&st->unresolved_count, 1u);
        b.ulist[so + pos] = i;
        b.ukeys[so + pos] = ~0ull;
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// Rings 2, 4, ... max_ring for the queries the first ring did not certify (dlist), one WAVE per query: the rows of the
// cell block go round the 64 lanes (cells of the previous block are skipped as in nn_ring), the lanes' results are merged
// with the explicit tie rule, and the certification test is the same.  What is still uncertified goes to the brute-force
// fallback list.  Fixed grid; the waves stride over the list.
__global__ __launch_bounds__(kNnThreads) void nn_ring_wide(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int count = (int)st->deferred_count;
  constexpr int kWaves = kNnThreads / 64;
  if ((int)blockIdx.x * kWaves >= count) return;
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t so = (size_t)pair * b.ns_cap;
  const uint2* words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const float4* tq = b.tq + (size_t)pair * b.nt_cap;
  const uint32_t* rowbits = b.have_rowbits ? b.rowbits + (size_t)pair * kMaxRowWords : nullptr;
  const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
  const float inv_h = st->inv_h;
  const float hh = st->h, slack = 2.0e-3f * hh;
  const int nx = st->nx, ny = st->ny, nz = st->nz, wx = st->wx;
  for (int e = (int)blockIdx.x * kWaves + (int)(threadIdx.x >> 6); e < count; e += (int)gridDim.x * kWaves) {   // wave-uniform
    const int i = b.dlist[so + e];
    double px, py, pz;
    transform_point(st->M, ld_src(b, so + i), px, py, pz);
    const float qx = (float)px, qy = (float)py, qz = (float)pz;
    const int cx = cell_coord(qx, ox, inv_h), cy = cell_coord(qy, oy, inv_h), cz = cell_coord(qz, oz, inv_h);
    Best best = {b.d2[so + i], b.idx[so + i], INFINITY};            // what the first ring found (uncertified)
    bool resolved = false;
    int rp = 1;
    for (int r = 2; r <= b.max_ring; r *= 2) {
      const int x0 = max(cx - r, 0), x1 = min(cx + r, nx - 1);
      const int y0 = max(cy - r, 0), y1 = min(cy + r, ny - 1);
      const int z0 = max(cz - r, 0), z1 = min(cz + r, nz - 1);
      if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
        const int nyr = y1 - y0 + 1, nrows = nyr * (z1 - z0 + 1);
        for (int k = lane; k < nrows; k += 64) {
          const int zr = k / nyr;
          const int z = z0 + zr, y = y0 + (k - zr * nyr);
          if (rowbits && !row_occupied(rowbits, ny, z, y)) continue;       // an empty row: one bit instead of two dependent loads
          // the first ring left an (uncertified) neighbour: only the part of the row inside the ball of that distance can hold a
          // closer point, or one that ties with it -- the other rows and cells are not looked up at all
          const float zl = oz + (float)z * hh, yl = oy + (float)y * hh;
          const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + hh)) - slack, 0.f);
          const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + hh)) - slack, 0.f);
          const float rest = best.d2 - fmaf(dy, dy, dz * dz);
          if (rest < 0.f) continue;
          int xa = x0, xb = x1;
          if (rest < 1.0e30f) {
            const float xr = sqrtf(rest) * 1.0001f + slack;
            xa = max(x0, cell_coord(qx - xr, ox, inv_h));
            xb = min(x1, cell_coord(qx + xr, ox, inv_h));
            if (xa > xb) continue;
          }
          const int rowbase = (z * ny + y) * wx;
          const bool inner = abs(y - cy) <= rp && abs(z - cz) <= rp;
          if (!inner) {
            search_row(words, cstart, tq, rowbase, xa, xb, qx, qy, qz, best);
          } else {
            const int xl1 = min(min(cx - rp - 1, nx - 1), xb), xr0 = max(max(cx + rp + 1, 0), xa);
            if (xa <= xl1) search_row(words, cstart, tq, rowbase, xa, xl1, qx, qy, qz, best);
            if (xr0 <= xb) search_row(words, cstart, tq, rowbase, xr0, xb, qx, qy, qz, best);
          }
        }
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float od = __shfl_xor(best.d2, off, 64);
        const int oj = __shfl_xor(best.j, off, 64);
        if (od < best.d2 || (od == best.d2 && oj >= 0 && (best.j < 0 || oj < best.j))) { best.d2 = od; best.j = oj; }
      }
      const float g = block_guarantee(st, qx, qy, qz, cx - r, cx + r, cy - r, cy + r, cz - r, cz + r);
      if (g == INFINITY || (g > 0.f && (best.d2 <= g * g || ring_irrelevant(b, g, best.d2)))) { resolved = true; break; }
      rp = r;
    }
    if (lane == 0) {
      b.d2[so + i] = best.d2;
      b.idx[so + i] = best.j;
      if (resolved) {
        const uint32_t key = __float_as_uint(best.d2);
        if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
      } else {
        const uint32_t pos = atomicAdd(&st->unresolved_count, 1u);
        b.ulist[so + pos] = i;
        b.ukeys[so + pos] = ~0ull;
      }
    }
  }
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// LDS-tiled exact brute force over every source point (SMHIP_NN_BRUTE, BASELINE config #2).
__global__ __launch_bounds__(kNnThreads) void nn_brute(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int count = st->ns;
  if ((int)(blockIdx.x * kNnThreads) >= count) return;
  __shared__ float4 s_t[kBruteTile];
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  const size_t so = (size_t)pair * b.ns_cap;
  const int i = blockIdx.x * kNnThreads + threadIdx.x;
  const bool active = i < count;
  float qx = 0, qy = 0, qz = 0;
  Best best = {INFINITY, -1, INFINITY};
  bool valid = false;
  if (active) {
    double px, py, pz;
    transform_point(st->M, ld_src(b, so + i), px, py, pz);
    qx = (float)px; qy = (float)py; qz = (float)pz;
    valid = isfinite(qx) && isfinite(qy) && isfinite(qz);
  }
  const float4* tq = b.tq + (size_t)pair * b.nt_cap;
  const int nt = st->nt;
  for (int base = 0; base < nt; base += kBruteTile) {
    const int m = min(kBruteTile, nt - base);
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += kNnThreads) s_t[k] = tq[base + k];
    __syncthreads();
    if (valid) {
      int k = 0;
#pragma unroll 4
      for (; k + 2 <= m; k += 2) test_ascending2(s_t[k], base + k, s_t[k + 1], base + k + 1, qx, qy, qz, best);
      if (k < m) test_ascending(s_t[k], base + k, qx, qy, qz, best);
    }
  }
  if (active) {
    b.d2[so + i] = best.d2;
    b.idx[so + i] = best.j;
    const uint32_t key = __float_as_uint(best.d2);
    if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
  }
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// Fallback for the few queries no ring could certify (far outside the target / very sparse).
// The work (U queries x nt targets) is spread over the whole chip: block (slice, pair) stages one
// slice of the target in LDS and sweeps ALL unresolved queries of the pair over it; the per-query
// winner is merged with a 64-bit atomicMin on (d2 bits << 32 | sorted position).
__device__ __forceinline__ void fallback_body(const IcpDev& b, PairState* st, int pair, int slice, int nslices, float4* s_t, uint32_t* s_last_p) {
  const int U = (int)st->unresolved_count;
  if (U == 0) return;                                     // the usual case: nothing reaches the fallback
  uint32_t& s_last = *s_last_p;
  const int nt = st->nt;
  const int per = (nt + nslices - 1) / nslices;
  const int lo = slice * per, hi = min(nt, lo + per);
  if (lo >= hi) return;
  const size_t so = (size_t)pair * b.ns_cap;
  const float4* tq = b.tq + (size_t)pair * b.nt_cap;
  for (int base = lo; base < hi; base += kBruteTile) {
    const int m = min(kBruteTile, hi - base);
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += kNnThreads) s_t[k] = tq[base + k];
    __syncthreads();
    for (int e = threadIdx.x; e < U; e += kNnThreads) {
      double px, py, pz;
      transform_point(st->M, ld_src(b, so + b.ulist[so + e]), px, py, pz);
      const float qx = (float)px, qy = (float)py, qz = (float)pz;
      Best best = {INFINITY, -1, INFINITY};
#pragma unroll 8
      for (int k = 0; k < m; ++k) test_ascending(s_t[k], base + k, qx, qy, qz, best);
      const unsigned long long key = ((unsigned long long)__float_as_uint(best.d2) << 32) | (uint32_t)best.j;
      atomicMin(&b.ukeys[so + e], key);
    }
  }
  // The slice that finishes last writes the results back (this used to be a second launch that was empty in nearly every
  // iteration).  Only this rare path pays for the agent-scope fences; the ticket orders the slices' atomicMin's before it.
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t active = (uint32_t)((nt + per - 1) / per);
    s_last = atomicAdd(&st->fallback_ticket, 1u) == active - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  for (int e = threadIdx.x; e < U; e += kNnThreads) {
    const int i = b.ulist[so + e];
    const unsigned long long key = __hip_atomic_load(&b.ukeys[so + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t dbits = (uint32_t)(key >> 32);
    b.d2[so + i] = __uint_as_float(dbits);
    st_match(b, so + i, (int)(uint32_t)(key & 0xffffffffu), 0.f);
    if (dbits < 0x7f800000u) atomicAdd(&b.hist[(size_t)pair * kHistBins + (dbits >> kHistShift)], 1u);
  }
  if (threadIdx.x == 0) st->fallback_ticket = 0;
}
__global__ __launch_bounds__(kNnThreads) void nn_fallback(IcpDev b) {
  const int pair = b.pair_base + blockIdx.y;
  PairState* st = &b.state[pair];
  if (st->done) return;
  __shared__ float4 s_t[kBruteTile];
  __shared__ uint32_t s_last;
  fallback_body(b, st, pair, (int)blockIdx.x, (int)gridDim.x, s_t, &s_last);
}


// ------------------------------------------------------------------------------------------
// K2/K3: quantile bin + point-to-plane accumulation
// ------------------------------------------------------------------------------------------
// rank of the quantile element: values.size() * quantile truncated (icp_fast.cc:86); quantile == 1
// takes the maximum (:82-84), i.e. rank n-1.
__device__ __forceinline__ int quantile_rank(uint32_t n_valid, float rho) {
  const double q = (double)rho;
  if (q >= 1.0) return (int)n_valid - 1;
  int k = (int)((double)n_valid * q);
  return min(k, (int)n_valid - 1);
}

// Finds the level-1 histogram bin holding the rank-k element.  All threads of a 256-thread block
// call it; results land in shared memory: s_out[0] = bin, s_out[1] = #elements below the bin,
// s_out[2] = n_valid, s_out[3] = k.
// Finds the level-1 histogram bin holding the rank-k element, from the counts of the eight bins each thread owns.  All threads of a
// 256-thread block call it; results land in shared memory: s_out[0] = bin, s_out[1] = #elements below the bin, s_out[2] = n_valid,
// s_out[3] = k.
__device__ __forceinline__ void find_quantile_bin_counts(const uint32_t* c, float rho, uint32_t* s_w, uint32_t* s_out) {
  constexpr int per = kHistBins / 256;
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < per; ++k) sum += c[k];
  uint32_t total;
  const uint32_t excl = block_excl_scan(sum, s_w, &total);
  if (threadIdx.x == 0) { s_out[0] = 0xffffffffu; s_out[1] = 0; s_out[2] = total; s_out[3] = 0; }
  __syncthreads();
  if (total > 0) {
    const uint32_t k = (uint32_t)quantile_rank(total, rho);
    if (excl <= k && k < excl + sum) {
      uint32_t run = excl;
#pragma unroll
      for (int t = 0; t < per; ++t) {
        if (k < run + c[t]) { s_out[0] = threadIdx.x * per + t; s_out[1] = run; s_out[3] = k; break; }
        run += c[t];
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void find_quantile_bin(const uint32_t* __restrict__ gh, float rho, uint32_t* s_w, uint32_t* s_out) {
  constexpr int per = kHistBins / 256;
  uint32_t c[per];
#pragma unroll
  for (int k = 0; k < per; ++k) c[k] = gh[threadIdx.x * per + k];
  find_quantile_bin_counts(c, rho, s_w, s_out);
}

// One block per pair, between nn_ball and the refinement kernels: does the quantile stay below
// every lower bound nn_ball recorded?  If not (or if every match must be exact) switch the ring
// search + fallback on for this iteration.
__device__ __forceinline__ void validate_bounds(const IcpDev& b, PairState* st, int pair, uint32_t* s_w, uint32_t* s_q) {
  find_quantile_bin(b.hist + (size_t)pair * kHistBins, b.rho, s_w, s_q);
  if (threadIdx.x == 0) {
    const bool any = st->hard_count > 0;
    // a bound sharing the quantile's bin is not provably above it: refine (conservative)
    const bool below = (st->min_lb_key >> kHistShift) <= s_q[0];
    st->refine = (any && (b.exact_all || below || s_q[2] == 0)) ? 1 : 0;
    if (st->refine) st->refine_total += 1;
    // fused path: every distance of the iteration is in the histogram now -- did the quantile's bin land in the band the fused
    // certificate pass was given?  Then its sums (below the band) and records (the band) are this iteration's; if not, or if
    // lower bounds must first be refined to matches, `accumulate` redoes them the plain way.
    st->spec_ok = (b.fused && st->band_lo > 0 && !st->refine && s_q[2] > 0 && (int)s_q[0] >= st->band_lo && (int)s_q[0] <= st->band_hi &&
                   st->deferred_count <= (uint32_t)kFusedListedMax) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void nn_validate(IcpDev b) {
  const int pair = b.pair_base + blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  validate_bounds(b, st, pair, s_w, s_q);
}

// The three refinement launches of an iteration (nn_validate, nn_ring<true>, nn_fallback) as ONE, a workgroup per pair, for
// launches of a few pairs: there every launch is ~5 us of dispatch whether it does anything or not, and in all but a poorly
// guessed first iteration the validation passes and nothing else happens.  When it does fail the one workgroup refines its
// pair's lower-bounded queries itself -- slower than the spread-out kernels, which is why batches and the modes that refine
// in every iteration (exact_matches, find_closests) keep those.  Same results: the same bodies, a different work split.
__global__ __launch_bounds__(kNnThreads) void nn_refine_one(IcpDev b) {
  const int pair = b.pair_base + blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  __shared__ uint32_t s_hist[kHistBins];
  __shared__ float4 s_t[kBruteTile];
  __shared__ uint32_t s_last;
  validate_bounds(b, st, pair, s_w, s_q);
  __syncthreads();
  if (!st->refine) return;                                 // (written by thread 0 above: visible after the barrier)
  ring_body<true>(b, st, pair, 0, 1, s_hist);
  __syncthreads();
  fallback_body(b, st, pair, 0, 1, s_t, &s_last);
}

// J = [p x n ; n], r = (p - q) . n ; acc += upper(J J^T), J r, 1     (icp_fast.cc:182-202, 256-303)
// p = the source point already moved (transform_point).  Every producer of sums -- accumulate, the fused certificate pass, the
// listed search's epilogue, finalize -- goes through this one function: a match contributes the same 28 doubles wherever it is met.
// Column 27 stays zero: the reference forms the sum of the kept distances' roots only in the iteration it leaves the loop with
// (icp_fast.cc:516-522), and so does the device -- final_score, one pass over the last iteration's distances -- instead of a root,
// a reciprocal and a Newton step per point and iteration (12 of the certificate pass's ~170 vector instructions per round).
// COUNT = false: the caller counts the kept matches itself (a wave's kept lanes by one scalar popcount instead of an f64 add per lane).
template <bool COUNT = true>
__device__ __forceinline__ void accumulate_terms_p(double px, double py, double pz, const float4 q4, const float4 n4, double* acc) {
  const double nx = n4.x, ny = n4.y, nz = n4.z;
  double J[6];
  J[0] = py * nz - pz * ny;
  J[1] = pz * nx - px * nz;
  J[2] = px * ny - py * nx;
  J[3] = nx; J[4] = ny; J[5] = nz;
  const double r = (px - (double)q4.x) * nx + (py - (double)q4.y) * ny + (pz - (double)q4.z) * nz;
  int c = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int e = a; e < 6; ++e) acc[c++] += J[a] * J[e];
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;     // b = -sum(J r): sign applied at solve time
  if (COUNT) acc[28] += 1.0;
}
__device__ __forceinline__ void accumulate_terms(const double* M, const float4 s4, const float4 q4, const float4 n4, double* acc) {
  double px, py, pz;
  transform_point(M, s4, px, py, pz);
  accumulate_terms_p(px, py, pz, q4, n4, acc);
}
// Block reduction of kAccCols-3 = 29 doubles; thread 0 ends up with the totals in acc[].
// v of the lanes a DPP control reads from, 0.0 where the control has no source lane or the row is masked off
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the wave, left in lane 63: an inclusive scan inside every row of 16 lanes (row_shr 1, 2, 4, 8), then the row
// totals carried across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3).  Register-to-register DPP moves:
// the ds_bpermute form of __shfl_down took 3.5 us for the 29 columns.
__device__ __forceinline__ double wave_sum_to_last(double v) {
  v += dpp_or_zero<0x111, 0xf>(v);
  v += dpp_or_zero<0x112, 0xf>(v);
  v += dpp_or_zero<0x114, 0xf>(v);
  v += dpp_or_zero<0x118, 0xf>(v);
  v += dpp_or_zero<0x142, 0xa>(v);
  v += dpp_or_zero<0x143, 0xc>(v);
  return v;
}

// Block reduction of kAccCols-3 = 29 doubles over 4 waves: s_out[c] = the totals (valid after the call for every thread).
__device__ __forceinline__ void block_reduce29(double* acc, double (*s_red)[29], double* s_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = wave_sum_to_last(acc[c]);
  if (lane == 63)
#pragma unroll
    for (int c = 0; c < 29; ++c) s_red[wave][c] = acc[c];
  __syncthreads();
  if (threadIdx.x < 29) s_out[threadIdx.x] = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
  __syncthreads();
}

// One block (kAccThreads * ITEMS source points) of a pair's ErrorElements / ComputePointToPlane sums with the quantile's bin known:
// row `blk` of partials, the bin's members as records in the block's four wave segments.
template <int ITEMS>
__device__ __forceinline__ void accumulate_block(const IcpDev& b, PairState* st, int pair, int blk, const double* Mc,
                                                 uint32_t* s_w, uint32_t* s_q, double (*s_red)[29], double* s_out) {
  const int ns = st->ns;
  const int base = blk * (kAccThreads * ITEMS);
  find_quantile_bin(b.hist + (size_t)pair * kHistBins, b.rho, s_w, s_q);
  const uint32_t qbin = s_q[0];
  double acc[29];
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = 0.0;
  const size_t so = (size_t)pair * b.ns_cap;
  // the three streamed values of a query (d2, source point, match) are loaded one round ahead of their use, without
  // looking at d2 first, so that their latency is not in series with the gathers of the matched target point and normal
  const size_t to = (size_t)pair * b.nt_cap;
  const int seg = blk * (kAccThreads / 64) + (int)(threadIdx.x >> 6);            // this wave's segment: 64 * ITEMS slots
  const size_t segbase = (size_t)pair * 2 * b.bl_stride + (size_t)seg * (64 * ITEMS);
  int wcount = 0;
  // two-deep software pipeline (as nn_certify): stream loads two rounds ahead, the gathers of the matched target point and
  // normal one round ahead
  int ic = min(base + (int)threadIdx.x, ns - 1);
  float d_1 = b.d2[so + ic];
  float4 s_1 = ld_src(b, so + ic);
  int j_1 = b.idx[so + ic];
  ic = min(base + kAccThreads + (int)threadIdx.x, ns - 1);
  float d_2 = b.d2[so + ic];
  float4 s_2 = ld_src(b, so + ic);
  int j_2 = b.idx[so + ic];
  // only a query below the quantile's bin uses its matched point and normal (the 30 % trimmed ones and the bin's own members do
  // not): its d2 arrived with its match id -- one round ahead of the gather -- so the gather is issued for those lanes alone
  float4 q_1 = make_float4(0, 0, 0, 0), n_1 = make_float4(0, 0, 0, 0);
  if ((__float_as_uint(d_1) >> kHistShift) < qbin) { q_1 = b.tq[to + max(j_1, 0)]; n_1 = b.tn[to + max(j_1, 0)]; }
#pragma unroll 2
  for (int it = 0; it < ITEMS; ++it) {
    const int i = base + it * kAccThreads + threadIdx.x;
    const float d = d_1;
    const float4 s4 = s_1;
    const float4 q4 = q_1, n4 = n_1;
    const int jm = j_1;
    d_1 = d_2; s_1 = s_2; j_1 = j_2;
    if (it + 1 < ITEMS && (__float_as_uint(d_1) >> kHistShift) < qbin) { q_1 = b.tq[to + max(j_1, 0)]; n_1 = b.tn[to + max(j_1, 0)]; }
    if (it + 2 < ITEMS) {
      ic = min(i + 2 * kAccThreads, ns - 1);
      d_2 = b.d2[so + ic];
      s_2 = ld_src(b, so + ic);
      j_2 = b.idx[so + ic];
    }
    bool boundary = false;
    if (i < ns) {
      const uint32_t key = __float_as_uint(d);
      if (key < 0x7f800000u) {
        const uint32_t bin = key >> kHistShift;
        if (bin < qbin) accumulate_terms(Mc, s4, q4, n4, acc);
        else boundary = bin == qbin;
      }
    }
    // the quantile bin's members are left for finalize as records (source point, d2, match: all it needs of them), compacted
    // per wave into the wave's own segment, in the order the wave meets them: no atomics, and an order that does not change
    // from run to run
    const unsigned long long bm = __ballot(boundary);
    if (boundary) {
      const size_t at = segbase + wcount + __popcll(bm & ((1ull << (threadIdx.x & 63)) - 1ull));
      b.rec_a[at] = make_float4(s4.x, s4.y, s4.z, d);
      b.rec_j[at] = jm;
    }
    wcount += (int)__popcll(bm);
  }
  if ((threadIdx.x & 63) == 0) b.gcount[(size_t)pair * b.seg_stride + seg] = (uint32_t)wcount;
  block_reduce29(acc, s_red, s_out);
  if (threadIdx.x < 29) b.partials[((size_t)pair * b.part_stride + blk) * kAccCols + threadIdx.x] = s_out[threadIdx.x];
}

template <int ITEMS>
__global__ __launch_bounds__(kAccThreads) void accumulate(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  if (b.fused && st->spec_ok) return;    // the fused certificate pass + the listed search's epilogue already hold this iteration's sums
  if (blk * (kAccThreads * ITEMS) >= st->ns) return;
  double Mc[12];                         // read before the first store: scalar loads, held in SGPRs (see nn_ball_lds)
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  __shared__ double s_red[4][29];
  __shared__ double s_out[29];
  accumulate_block<ITEMS>(b, st, pair, blk, Mc, s_w, s_q, s_red, s_out);
}


// ------------------------------------------------------------------------------------------
// The fused steady-state iteration: certificate pass + ErrorElements / ComputePointToPlane sums in ONE pass over the source
// (icp_fast.cc:484-523 per iteration: FindClosests for the queries whose match provably has not changed, then the normal
// equations over the matches below the trimming quantile).
//
// nn_certify followed by `accumulate` streams every source point twice per iteration (12 B point + 4 B bound + 4 B match in,
// 4 B distance out; then distance, point and match in again).  What stands between them is the quantile: which matches are
// summed is only known once every distance is.  But the quantile barely moves once ICP has settled, so finalize predicts the
// histogram bins it can fall in next ([band_lo, band_hi], 1-3 bins of 2048) and this pass, which has point, match, matched
// target point and distance in registers anyway, sums the certified matches below the band on the spot, drops those above
// it, and leaves the band's members as records (source point, distance, match id) for finalize to select among.  The queries
// whose certificate fails go to the wave's own segment of dlist; the listed search (nn_ball_listed_items) finds their matches
// and records those inside the band, iteration_sums adds those below it.  nn_validate then checks the prediction against the
// completed histogram (PairState::spec_ok); a miss costs one plain `accumulate` for that pair.  No atomics with a return value
// inside the loop (the lists are per-wave segments, counted in SGPRs), so the two-deep load pipeline never drains.
// NABO = true, the reference-search form: the traversal certificates of the libnabo walk instead of the distance bounds, the
// failing queries to the walk's four class lists (once per wave, after the loop), and after the walk nabo_validate +
// accumulate_listed in the roles of nn_validate and of iteration_sums' listed matches.
// Exactness: the same matches, the same distances, the same kept set as the separate passes; only the order in which the
// 29 sums are added differs (1e-16 relative), and it is a fixed order -- the result is reproducible bit for bit.
// number of set bits of `mask` below this lane (v_mbcnt_lo / _hi: two instructions, no lane mask to build and AND)
__device__ __forceinline__ uint32_t rank_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// segbase: wave-uniform (the stores take the scalar base + 32-bit lane offset form)
// bm: the wave's ballot of `band`
__device__ __forceinline__ void emit_record(const IcpDev& b, size_t segbase, int& wcount, bool band, unsigned long long bm, const float4 s, float d, int j) {
  if (band) {
    const uint32_t k = rank_below(bm);
    char* __restrict__ ra = reinterpret_cast<char*>(b.rec_a + segbase + wcount);
    char* __restrict__ rj = reinterpret_cast<char*>(b.rec_j + segbase + wcount);
    *reinterpret_cast<float4*>(ra + k * 16u) = make_float4(s.x, s.y, s.z, d);
    *reinterpret_cast<int32_t*>(rj + k * 4u) = j;
  }
  wcount += (int)__popcll(bm);
}

// SHADOW = true: the bound and the match come from their 4-byte shadow (IcpDev::mb): 16 streamed bytes per point in instead of 20
// (launches whose targets all have fewer than 32 767 points).  The shadow's bound is never larger than the recorded one, so a few
// more certificates fail and are searched -- the matches, distances and kept set are the same.
template <int ITEMS, bool NABO = false, bool SHADOW = false>
__global__ __launch_bounds__(kNnThreads, NABO ? 4 : 1) void nn_certify_acc(IcpDev b, int nblk) {   // (NABO: 129 registers without the hint, one wave per SIMD less)
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int ns = st->ns;
  const int base = blk * (kNnThreads * ITEMS);
  if (base >= ns) return;
  double Mc[12];                         // read before the first store: scalar loads, held in SGPRs (see nn_ball_lds)
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  // ... but the translation in VGPRs: fma(M[2], z, M[3]) with both M's in SGPRs is a move into a VGPR pair + v_fmac per coordinate
  // and point (one scalar operand per instruction); with the addend in a VGPR it is the one v_fma_f64
  asm volatile("" : "+v"(Mc[3]), "+v"(Mc[7]), "+v"(Mc[11]));
  const Pot pot = {(float)st->pot_a, (float)st->pot_b, 0.f, 0.f};
  const int band_lo = st->band_lo, band_hi = st->band_hi;      // band_lo = 0, band_hi = -1: no prediction, nothing summed or recorded
  __shared__ uint32_t s_hist[kHistBins];
  __shared__ double s_red[4][29];
  __shared__ double s_out[29];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t so = (size_t)pair * b.ns_cap, to = (size_t)pair * b.nt_cap;
  // the pair's arrays as uniform byte bases + 32-bit offsets (a pair's arrays are < 4 GB): the loads and stores take the scalar
  // base + vector offset form instead of a 64-bit address per lane and access (a sixth of the loop's vector instructions)
  const char* __restrict__ tqb = reinterpret_cast<const char*>(b.tq + to);
  const char* __restrict__ tnb = reinterpret_cast<const char*>(b.tn + to);
  const char* __restrict__ srcb = reinterpret_cast<const char*>(b.src3 + 3 * so);
  const char* __restrict__ lbb = reinterpret_cast<const char*>(b.lb + so);
  const char* __restrict__ idxb = reinterpret_cast<const char*>(b.idx + so);
  const char* __restrict__ mbb = reinterpret_cast<const char*>(b.mb + so);
  char* __restrict__ d2b = reinterpret_cast<char*>(b.d2 + so);
  auto ld_s = [&](int k) { const float3 v = *reinterpret_cast<const float3*>(srcb + __umul24((uint32_t)k, 12u)); return make_float4(v.x, v.y, v.z, 0.f); };   // (k < 2^24; the 32-bit multiply is a quarter-rate instruction)
  auto ld_l = [&](int k) { return *reinterpret_cast<const float*>(lbb + (uint32_t)k * 4u); };
  auto ld_j = [&](int k) { return *reinterpret_cast<const int*>(idxb + (uint32_t)k * 4u); };
  auto ld_m = [&](int k) { return *reinterpret_cast<const uint32_t*>(mbb + (uint32_t)k * 4u); };
  auto ld_t = [&](const char* base, int j) { const float3 v = *reinterpret_cast<const float3*>(base + (uint32_t)max(j, 0) * 16u); return make_float4(v.x, v.y, v.z, 0.f); };
  auto st_d = [&](int k, float v) { *reinterpret_cast<float*>(d2b + (uint32_t)k * 4u) = v; };
  const float r_need = 0.9f * sqrtf(st->rcap2);     // a hard query's bound must stay well above the quantile
  uint32_t min_lb = 0xffffffffu;
  // this wave's segments: 64 * ITEMS slots each (the wave's number read into an SGPR: the segment bases below are scalars)
  const int seg = blk * (kNnThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t recbase = (size_t)pair * 2 * b.bl_stride + (size_t)seg * (64 * ITEMS);
  char* __restrict__ dsegb = reinterpret_cast<char*>(b.dlist + (size_t)pair * b.dl_stride + (size_t)seg * (64 * ITEMS));
  int nrec = 0, ndef = 0, nkept = 0;
  uint32_t nabo_failed = 0;
  double acc[29];
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = 0.0;
  // two-deep software pipeline: a round's streamed values (point, bound, previous match id) are loaded two rounds ahead, the
  // gathers of the previous match and its normal one round ahead (for every lane: seven in ten use both)
  int ic = min(base + (int)threadIdx.x, ns - 1);
  float4 s_1 = ld_s(ic);
  float l_1, l_2;
  int j_1, j_2;
  uint32_t m_2 = 0;                                  // SHADOW: a round's packed (bound, match), unpacked when it becomes round 1
  if (SHADOW) { const uint32_t m = ld_m(ic); l_1 = shadow_bound(m); j_1 = shadow_match(m); }
  else { l_1 = ld_l(ic); j_1 = ld_j(ic); }
  ic = min(base + kNnThreads + (int)threadIdx.x, ns - 1);
  float4 s_2 = ld_s(ic);
  if (SHADOW) { m_2 = ld_m(ic); l_2 = 0.f; j_2 = 0; }
  else { l_2 = ld_l(ic); j_2 = ld_j(ic); }
  float4 t_1 = ld_t(tqb, j_1);
  float4 n_1 = ld_t(tnb, j_1);
  // One round of 256 points.  GATHER / STREAM: whether the round still issues the gathers of the next round and the stream loads of
  // the round after it -- compile-time flags (the last two rounds are written out after the loop), so that inside the loop the
  // pipeline registers are plainly overwritten: as run-time conditions every guarded load kept its old value alive and cost a
  // register copy per round.
  auto round = [&](const int it, auto GATHER, auto STREAM) {
    const int i = base + it * kNnThreads + threadIdx.x;
    bool hard = false, fail = false, band = false;
    const float4 s = s_1;
    const float l = l_1;
    const int j = j_1;
    const float4 t = t_1, n = n_1;
    s_1 = s_2;
    if (SHADOW) { l_1 = shadow_bound(m_2); j_1 = shadow_match(m_2); }
    else { l_1 = l_2; j_1 = j_2; }
    // (the normals are gathered in the one iteration without a prediction too: a branch around the load costs its register copies
    // in every other)
    if (GATHER.value) { t_1 = ld_t(tqb, j_1); n_1 = ld_t(tnb, j_1); }
    if (STREAM.value) {
      ic = min(i + 2 * kNnThreads, ns - 1);
      s_2 = ld_s(ic);
      if (SHADOW) m_2 = ld_m(ic);
      else { l_2 = ld_l(ic); j_2 = ld_j(ic); }
    }
    float d1 = 0.f;
    unsigned long long m_fail = 0, m_band = 0, m_hard = 0;    // the wave's ballots of fail / band / hard
    if (NABO) {
      bool keep = false;
      if (i < ns) {
        double px, py, pz;
        transform_point(Mc, s, px, py, pz);
        const float qx = (float)px, qy = (float)py, qz = (float)pz;
        fail = true;
        // the traversal certificate of the libnabo walk (nn_certify<., true>, with its expression and its |s|): the query keeps
        // its id -- by the same walk -- and only its distance to that id is recomputed
        const float Pn = pot_at(pot, norm3(s.x, s.y, s.z));
        if (isfinite(qx) && isfinite(qy) && isfinite(qz) && l > 0.f && j >= 0 &&
            l - Pn - 4.0e-7f * (l + Pn) - 1.3e-7f * (fabsf(qx) + fabsf(qy) + fabsf(qz)) > 0.f) {
          d1 = dist2(t, qx, qy, qz);               // the bucket scan's arithmetic: the bits the walk would produce
          st_d(i, d1);
          const int bin = (int)(__float_as_uint(d1) >> kHistShift);
          atomicAdd(&s_hist[bin], 1u);
          fail = false;
          if (bin < band_lo) { accumulate_terms_p<false>(px, py, pz, t, n, acc); keep = true; }
          else band = bin <= band_hi;
        }
      }
      nkept += (int)__popcll(__ballot(keep));        // (counted where the whole wave is: an update inside the branches would be per lane)
    } else {
      // Every predicate of the round as a WAVE MASK: the ballots of the single compares (a compare's own scalar result) combined
      // by scalar ANDs, over values all lanes have (the loads are clamped to the pair's last point); a lane's own predicate is
      // its bit of the mask (inverse ballot: no instruction).  As nested branches each of fail / band / hard was a boolean carried
      // through the branches in a VGPR and compared again for its ballot, and the kept count an add per lane.
      const auto B = [](bool p) { return (unsigned long long)__builtin_amdgcn_ballot_w64(p); };
      const auto P = [](unsigned long long m) { return (bool)__builtin_amdgcn_inverse_ballot_w64(m); };
      double px, py, pz;
      transform_point(Mc, s, px, py, pz);
      const float qx = (float)px, qy = (float)py, qz = (float)pz;
      // as nn_certify; |s| by the hardware root (1 ulp: bound_now's slack of 1e-4 of the potential covers a million of those)
      const float Lp = bound_now(l, pot_at(pot, __builtin_amdgcn_sqrtf(fmaf(s.z, s.z, fmaf(s.y, s.y, s.x * s.x)))));
      const unsigned long long m_inr = B(i < ns);
      const unsigned long long m_pos = m_inr & B(Lp > 0.f);
      const unsigned long long m_has = B(l > 0.f) & B(j >= 0);
      d1 = dist2(t, qx, qy, qz);
      // still the unique nearest neighbour: exact, no search.  (A query with a non-finite coordinate has a non-finite d1: the
      // compare fails, as the explicit test did.)
      const unsigned long long m_ok = m_pos & m_has & B(d1 < Lp * Lp);
      // still provably farther than the trimming radius.  All three coordinates finite <=> 0 * qx + 0 * qy + 0 * qz == 0 (inf and NaN
      // give NaN): four instructions; isfinite is a class compare whose ballot costs a select and a compare more, per coordinate
      const float zq = fmaf(qz, 0.f, fmaf(qy, 0.f, qx * 0.f));
      m_hard = m_pos & B(zq == 0.f) & ~m_has & B(l < 0.f) & B(Lp >= r_need);
      m_fail = m_inr & ~(m_ok | m_hard);
      const int bin = (int)(__float_as_uint(d1) >> kHistShift);
      const unsigned long long m_keep = m_ok & B(bin < band_lo);          // below every bin the quantile can fall in: kept
      m_band = m_ok & ~m_keep & B(bin <= band_hi);                        // the quantile decides: finalize
      nkept += (int)__popcll(m_keep);                                     // (a scalar count, no f64 add per lane)
      fail = P(m_fail); band = P(m_band); hard = P(m_hard);
      if (P(m_ok)) {
        st_d(i, d1);
        atomicAdd(&s_hist[bin], 1u);
      }
      if (P(m_keep)) accumulate_terms_p<false>(px, py, pz, t, n, acc);
      if (hard) {
        const float lb2 = Lp * Lp;
        st_d(i, lb2);
        atomicAdd(&s_hist[__float_as_uint(lb2) >> kHistShift], 1u);
        min_lb = min(min_lb, __float_as_uint(lb2));
      }
    }
    if (NABO) {
      if (fail) nabo_failed |= 1u << it;             // to the class lists after the loop (nabo_list_append)
      m_band = __ballot(band);
    } else {   // failing certificates: the wave's own segment of dlist, in query order
      if (fail) *reinterpret_cast<int*>(dsegb + ((uint32_t)ndef + rank_below(m_fail)) * 4u) = i;
      ndef += (int)__popcll(m_fail);
    }
    emit_record(b, recbase, nrec, band, m_band, s, d1, j);
    const unsigned long long hm = m_hard;
    if (hm) {                                       // rare: lower-bounded queries keep their global list
      uint32_t basepos = 0;
      if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
      basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
      if (hard) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = i;
    }
  };
  static_assert(ITEMS >= 2, "the pipeline's last two rounds are written out");
#pragma unroll 2
  for (int it = 0; it < ITEMS - 2; ++it) round(it, std::true_type{}, std::true_type{});
  round(ITEMS - 2, std::true_type{}, std::false_type{});
  round(ITEMS - 1, std::false_type{}, std::false_type{});
  if (lane == 0) {
    b.gcount[(size_t)pair * b.seg_stride + seg] = (uint32_t)nrec;
    b.dcount[(size_t)pair * b.seg_stride + seg] = ndef;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) min_lb = min(min_lb, (uint32_t)__shfl_xor((int)min_lb, off, 64));
  if (lane == 0 && min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
  if (lane == 0) acc[28] = (double)nkept;           // the wave's kept matches (counted in an SGPR)
  block_reduce29(acc, s_red, s_out);                // (its barriers also order the histogram updates before the flush)
  if (threadIdx.x < 29) b.partials[((size_t)pair * b.part_stride + blk) * kAccCols + threadIdx.x] = s_out[threadIdx.x];
  if (NABO) nabo_list_append<ITEMS>(b, st, so, base, ns, nabo_failed, lane);      // (after the sums: their 58 registers are free again)
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// The search of the queries whose certificate failed (a few thousand per pair once the pose has settled, most of them in the
// first iterations after the switch), with up to 16 lanes per query: the rows of the query's ball go round the lanes, the lanes'
// results are merged with the sweep's tie rule.  One chunk of kNnThreads >> logL queries per call; `hard` / `band` / the record
// fields come back for the caller's list handling.
struct ListedCtx {
  const uint2* __restrict__ words; const uint32_t* __restrict__ cstart; const float4* __restrict__ tq;
  float ox, oy, oz, h, inv_h, r2cap;
  int nx, ny, nz, wx;
  uint32_t nt_last;
  bool have_prev;
  Pot pot;
};
__device__ __forceinline__ ListedCtx listed_ctx(const IcpDev& b, const PairState* st, int pair) {
  ListedCtx c;
  c.words = b.words + (size_t)pair * kMaxGridWords;
  c.cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  c.tq = b.tq + (size_t)pair * b.nt_cap;
  c.ox = st->origin[0]; c.oy = st->origin[1]; c.oz = st->origin[2];
  c.h = st->h; c.inv_h = st->inv_h; c.r2cap = st->rcap2;
  c.nx = st->nx; c.ny = st->ny; c.nz = st->nz; c.wx = st->wx;
  c.nt_last = (uint32_t)max(st->nt - 1, 0);
  c.have_prev = st->iter > 0;
  c.pot = {(float)st->pot_a, (float)st->pot_b, 0.f, 0.f};
  return c;
}
// query i (or none: i < 0) searched by the L = 1 << logL lanes of its group; lane sub == 0 of the group stores the result
__device__ __forceinline__ void listed_search_one(const IcpDev& b, const PairState* st, const ListedCtx& c, size_t so, int i, int sub, int L,
                                                  uint32_t* s_hist, uint32_t& min_lb, bool& hard, bool track_band, int band_lo, int band_hi,
                                                  bool& band, float4& srec, float& drec, int& jrec) {
  if (i < 0) {
    // (the butterfly below is executed by whole groups: a group without a query has nothing to merge, but lanes of other groups
    // in the wave do -- __shfl_xor inside aligned groups never crosses into this one)
    return;
  }
  const float4 s4 = ld_src(b, so + i);
  double px, py, pz;
  transform_point(st->M, s4, px, py, pz);
  const float qx = (float)px, qy = (float)py, qz = (float)pz;
  Best best = {INFINITY, -1, INFINITY};
  const bool finite = isfinite(qx) && isfinite(qy) && isfinite(qz);
  float R2 = c.r2cap;
  int jp = -1;
  if (finite) {
    if (c.have_prev) {
      jp = b.idx[so + i];
      if (jp >= 0) {
        double ux, uy, uz;
        transform_point(st->M_prev, s4, ux, uy, uz);
        const float ex = qx - (float)ux, ey = qy - (float)uy, ez = qz - (float)uz;
        R2 = search_radius2(c.r2cap, dist2(c.tq[jp], qx, qy, qz), search_margin(sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)))));
      }
    }
    // every target point within sqrt(R2) of q lies in a cell meeting [q - Rs, q + Rs]^3
    const float Rs = sqrtf(R2) * 1.0001f + 1.0e-3f * c.h;
    const int x0 = max(cell_coord(qx - Rs, c.ox, c.inv_h), 0), x1 = min(cell_coord(qx + Rs, c.ox, c.inv_h), c.nx - 1);
    const int y0 = max(cell_coord(qy - Rs, c.oy, c.inv_h), 0), y1 = min(cell_coord(qy + Rs, c.oy, c.inv_h), c.ny - 1);
    const int z0 = max(cell_coord(qz - Rs, c.oz, c.inv_h), 0), z1 = min(cell_coord(qz + Rs, c.oz, c.inv_h), c.nz - 1);
    const int nyr = y1 - y0 + 1;
    const int nrows = (x0 <= x1 && y0 <= y1 && z0 <= z1) ? nyr * (z1 - z0 + 1) : 0;
    const float slack = 2.0e-3f * c.h;
    // This lane's rows of the ball, FOUR at a time with their lookups issued level by level: the four rows' words, then their run
    // bounds, then their first points.  A row at a time the walk is a chain of three dependent loads per row (and a listed query
    // has nothing else to do while it waits): the launches of the first iterations after the switch, with 10 000 and more queries
    // per pair, were 1.0 / 0.5 / 0.27 ms per 512 pairs for what is 3-4 rows and a handful of candidates per query.  The tests stay
    // in the sweep's order (rows ascending, a run's points ascending).
    const float inv_nyr = 1.0f / (float)max(nyr, 1);
    const uint32_t wa_i = (uint32_t)(x0 >> 5), wc_i = (uint32_t)(x1 >> 5);
    const uint32_t ma = (1u << (x0 & 31)) - 1u, mc = 0xffffffffu >> (31 - (x1 & 31));
    for (int rb = sub; rb < nrows; rb += 4 * L) {
      bool ok[4];
      uint2 wa[4], wc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rb + k * L, rr = min(r, nrows - 1);
        const int zr = (int)(((float)rr + 0.5f) * inv_nyr);   // rr / nyr (at most 29 x 29 rows: exact via float)
        const int z = z0 + zr, y = y0 + (rr - zr * nyr);
        const float zl = c.oz + (float)z * c.h, yl = c.oy + (float)y * c.h;
        const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + c.h)) - slack, 0.f);
        const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + c.h)) - slack, 0.f);
        ok[k] = r < nrows && !(fmaf(dy, dy, dz * dz) > R2);   // (a row outside the ball is skipped)
        const int rowbase = (z * c.ny + y) * c.wx;
        wa[k] = c.words[rowbase + wa_i];
        wc[k] = c.words[rowbase + wc_i];
      }
      uint32_t j0[4], j1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t sb = wa[k].y + __popc(wa[k].x & ma), se = wc[k].y + __popc(wc[k].x & mc);
        const bool has = ok[k] && se > sb;
        j0[k] = c.cstart[has ? sb : 0u];
        j1[k] = c.cstart[has ? se : 0u];                      // (no cell of the row in range: an empty run)
      }
      float4 t0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t0[k] = c.tq[min(j0[k], c.nt_last)];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (j1[k] > j0[k]) {
          test_ascending_ru(t0[k], (int)j0[k], qx, qy, qz, best);
          for (uint32_t j = j0[k] + 1; j < j1[k]; j += 4) {   // the rest of the run: four loads in flight, not one after the other
            const uint32_t e = j1[k] - 1u;
            const float4 a0 = c.tq[j], a1 = c.tq[min(j + 1u, e)], a2 = c.tq[min(j + 2u, e)], a3 = c.tq[min(j + 3u, e)];
            test_ascending_ru(a0, (int)j, qx, qy, qz, best);
            if (j + 1u <= e) test_ascending_ru(a1, (int)j + 1, qx, qy, qz, best);
            if (j + 2u <= e) test_ascending_ru(a2, (int)j + 2, qx, qy, qz, best);
            if (j + 3u <= e) test_ascending_ru(a3, (int)j + 3, qx, qy, qz, best);
          }
        }
      }
    }
  }
  // merge the L lanes of the query (butterfly inside aligned groups of L lanes; every lane ends with the result)
  for (int off = 1; off < L; off <<= 1) {
    const float od = __shfl_xor(best.d2, off, 64), os = __shfl_xor(best.s2, off, 64);
    const int oj = __shfl_xor(best.j, off, 64);
    best.s2 = fminf(fminf(best.s2, os), fmaxf(best.d2, od));
    if (od < best.d2 || (od == best.d2 && (unsigned)oj < (unsigned)best.j)) { best.d2 = od; best.j = oj; }
  }
  if (sub == 0) {
    float d2out = INFINITY, lbout = 0.f;
    int jout = -1;
    if (finite) {
      if (best.d2 <= R2) {                  // exact: everything within sqrt(R2) was seen
        d2out = best.d2;
        jout = best.j;
        lbout = sqrtf(fminf(best.s2, R2));  // every other point is at least this far
        if (track_band) {                   // in the predicted quantile band: a record for finalize's select
          const int bin = (int)(__float_as_uint(d2out) >> kHistShift);
          band = bin >= band_lo && bin <= band_hi;
          srec = s4; drec = d2out; jrec = jout;
        }
      } else {                              // certified lower bound: nothing lies within sqrt(R2)
        d2out = R2;
        jout = best.j >= 0 ? best.j : jp;   // an upper-bound seed for later iterations
        lbout = -sqrtf(R2);
        hard = true;
        min_lb = min(min_lb, __float_as_uint(R2));
      }
    }
    b.d2[so + i] = d2out;
    st_match(b, so + i, jout, with_pot(lbout, pot_at(c.pot, norm3(s4.x, s4.y, s4.z))));
    const uint32_t key = __float_as_uint(d2out);
    if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
  }
}
// lanes per query for a list of `count` entries: as many (a power of two, at most 16) as keep `budget` lanes busy
__device__ __forceinline__ int listed_lanes_log2(int count, int budget = kListedBlocks * kNnThreads) {
  int L = 1, logL = 0;
  while (L < 16 && 2 * L * count <= budget) { L *= 2; ++logL; }
  return logL;
}
__device__ __forceinline__ void listed_append_hard(const IcpDev& b, PairState* st, size_t so, bool hard, int i, int lane) {
  const unsigned long long hm = __ballot(hard);
  if (hm) {
    uint32_t basepos = 0;
    if (lane == 0) basepos = atomicAdd(&st->hard_count, (uint32_t)__popcll(hm));
    basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);   // lane 0 did the atomic
    if (hard) b.hlist[so + basepos + __popcll(hm & ((1ull << lane) - 1ull))] = i;
  }
}

// From the compacted list nn_certify filled with one atomic per wave (dlist, deferred_count): kListedBlocks workgroups per pair
// stride over it.
__global__ __launch_bounds__(kNnThreads, 6) void nn_ball_listed(IcpDev b, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int count = (int)st->deferred_count;
  if (count == 0) return;
  const int logL = listed_lanes_log2(count), L = 1 << logL;
  const int qpb = kNnThreads >> logL;                      // queries per workgroup and pass
  if (blk * qpb >= count) return;
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = (int)threadIdx.x & (L - 1), ql = (int)threadIdx.x >> logL;
  const size_t so = (size_t)pair * b.ns_cap;
  const ListedCtx c = listed_ctx(b, st, pair);
  uint32_t min_lb = 0xffffffffu;
  for (int base = blk * qpb; base < count; base += nblk * qpb) {        // workgroup-uniform
    const int e = base + ql;
    bool hard = false, band = false;
    float4 srec; float drec; int jrec;
    const int i = e < count ? b.dlist[so + e] : -1;
    listed_search_one(b, st, c, so, i, sub, L, s_hist, min_lb, hard, false, 0, -1, band, srec, drec, jrec);
    listed_append_hard(b, st, so, hard, i, lane);
  }
  if (min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&gh[k], v);
  }
}

// Fused path.  The fused certificate pass leaves the failing queries in per-wave segments of dlist (dcount).  Their number differs
// wildly from pair to pair -- in one iteration of the bench batch a median pair lists 2 500 queries and the worst 16 000; with
// half the guesses poor the spread is 100 to 117 000 -- and a launch that gives every pair the same workgroups lasts as long as
// its worst pair.  So: listed_plan (a workgroup per pair) turns the segment counts into offsets, totals the list and cuts it into
// ITEMS of one pass each (kNnThreads >> logL consecutive entries, logL from the list's length); nn_ball_listed_items is a
// fixed grid of workgroups that takes the items of ALL pairs of the launch in turn.  A match found here whose distance falls
// in the predicted quantile band becomes a record in its item's segment of region 1 (compacted inside the workgroup, in list
// order); the matches below the band are summed by iteration_sums, which walks the list once more.
__global__ __launch_bounds__(256) void listed_plan(IcpDev b) {
  const int pair = b.pair_base + blockIdx.x;
  PairState* st = &b.state[pair];
  __shared__ uint32_t s_w[17];
  int32_t* dc = b.dcount + (size_t)pair * b.seg_stride;
  const int nseg0 = ((st->ns + kNnThreads * kCertifyItems - 1) / (kNnThreads * kCertifyItems)) * (kNnThreads / 64);
  const int per = (nseg0 + 255) >> 8;                      // <= kFinalizeMaxSeg / 256
  const int s0 = (int)threadIdx.x * per;
  uint32_t c[kFinalizeMaxSeg / 256];
  uint32_t mine = 0;
  const bool live = !st->done;                              // (a finished pair's certificate pass did not run: its counts are stale)
#pragma unroll
  for (int k = 0; k < kFinalizeMaxSeg / 256; ++k) {
    c[k] = (live && k < per && s0 + k < nseg0) ? (uint32_t)dc[s0 + k] : 0u;
    mine += c[k];
  }
  uint32_t total;
  uint32_t o = block_excl_scan(mine, s_w, &total);
#pragma unroll
  for (int k = 0; k < kFinalizeMaxSeg / 256; ++k)
    if (k < per && s0 + k < nseg0) { dc[s0 + k] = (int32_t)o; o += c[k]; }          // counts -> exclusive offsets, in place
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) st->listed_ticket = 0;
    dc[nseg0] = (int32_t)total;
    st->deferred_count = total;
    const int qpb = kNnThreads >> listed_lanes_log2((int)total, b.listed_lane_budget);
    b.litems[pair] = total ? (total + qpb - 1) / qpb : 0u;
  }
}

__global__ __launch_bounds__(kNnThreads, 4) void nn_ball_listed_items(IcpDev b) {
  __shared__ uint32_t s_hist[kHistBins];
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_plan[kListedMaxPairs + 1];         // items before pair p of the launch
  __shared__ uint32_t s_off[2 * kFinalizeMaxSeg];          // the current pair: entries before each segment of its list
  __shared__ uint32_t s_wc[2][kNnThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int np = b.npairs;
  {
    constexpr int kPer = kListedMaxPairs / kNnThreads;
    const int per = (np + kNnThreads - 1) / kNnThreads;
    const int p0 = (int)threadIdx.x * per;
    uint32_t v[kPer], mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { v[k] = (k < per && p0 + k < np) ? b.litems[b.pair_base + p0 + k] : 0u; mine += v[k]; }
    uint32_t total;
    uint32_t o = block_excl_scan(mine, s_w, &total);
#pragma unroll
    for (int k = 0; k < kPer; ++k) if (k < per && p0 + k < np) { s_plan[p0 + k] = o; o += v[k]; }
    if (threadIdx.x == 0) s_plan[np] = total;
    __syncthreads();
  }
  // The items go to the workgroups a few at a time (a ticket counter in the launch's first pair: an item's cost depends on how
  // many rows its queries' balls cross, so equal runs fixed in advance end unevenly); consecutive items mostly belong to one pair.
  const int nitems = (int)s_plan[np];
  __shared__ uint32_t s_ticket;
  const int grain = b.listed_grain;
  int it0, it1;
  if (grain > 0) {
    it0 = it1 = 0;
  } else {
    const int nwg = (int)gridDim.x;
    const int vb = ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3);     // gridDim.x is a multiple of 8
    it0 = (int)(((long long)vb * nitems) / nwg); it1 = (int)(((long long)(vb + 1) * nitems) / nwg);
    if (it0 >= it1) return;
  }
  int lo = 0;
  int cur = -1, count = 0, logL = 0, nseg0 = 0, top = 1, band_lo = 0, band_hi = -1;
  uint32_t min_lb = 0xffffffffu;
  ListedCtx c;
  PairState* st = nullptr;
  auto flush = [&]() {                                     // the finished pair's histogram and smallest lower bound
    if (min_lb != 0xffffffffu) atomicMin(&st->min_lb_key, min_lb);
    min_lb = 0xffffffffu;
    __syncthreads();
    uint32_t* gh = b.hist + (size_t)(b.pair_base + cur) * kHistBins;
    for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
      const uint32_t v = s_hist[k];
      if (v) atomicAdd(&gh[k], v);
    }
    __syncthreads();
  };
  for (int item = it0;; ++item) {                          // workgroup-uniform
    if (item >= it1) {
      if (grain <= 0) break;
      __syncthreads();
      if (threadIdx.x == 0) s_ticket = atomicAdd(&b.state[b.pair_base].listed_ticket, (uint32_t)grain);
      __syncthreads();
      it0 = (int)min(s_ticket, (uint32_t)nitems);
      it1 = min(it0 + grain, nitems);
      if (it0 >= it1) break;
      item = it0;
      int hi = np - 1;                                     // the last pair with s_plan[p] <= it0 (pairs without items: equal entries)
      lo = 0;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_plan[mid] <= (uint32_t)it0) lo = mid; else hi = mid - 1; }
    }
    while (s_plan[lo + 1] <= (uint32_t)item) ++lo;
    if (lo != cur) {
      if (cur >= 0) flush();
      cur = lo;
      const int pair = b.pair_base + cur;
      st = &b.state[pair];
      const int32_t* __restrict__ dc = b.dcount + (size_t)pair * b.seg_stride;
      nseg0 = ((st->ns + kNnThreads * kCertifyItems - 1) / (kNnThreads * kCertifyItems)) * (kNnThreads / 64);
      count = dc[nseg0];
      logL = listed_lanes_log2(count, b.listed_lane_budget);
      for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
      top = 1;
      while (2 * top < nseg0) top *= 2;
      for (int x = threadIdx.x; x < 2 * top; x += kNnThreads) s_off[x] = x < nseg0 ? (uint32_t)dc[x] : 0xffffffffu;
      c = listed_ctx(b, st, pair);
      // (a list beyond kFusedListedMax: nn_validate drops the speculation and accumulate makes the records)
      band_lo = count <= kFusedListedMax ? st->band_lo : 0;
      band_hi = st->band_hi;
      __syncthreads();
    }
    const int pair = b.pair_base + cur;
    const int chunk = item - (int)s_plan[cur];
    const int L = 1 << logL, qpb = kNnThreads >> logL;
    const int sub = (int)threadIdx.x & (L - 1), ql = (int)threadIdx.x >> logL;
    const size_t so = (size_t)pair * b.ns_cap;
    const int e = chunk * qpb + ql;
    int i = -1;
    if (e < count) {                                       // entry e = the (e - s_off[seg])-th of the last segment with s_off[seg] <= e
      const int32_t* __restrict__ dl = b.dlist + (size_t)pair * b.dl_stride;
      int sg = 0;
      for (int step = top; step > 0; step >>= 1) { const int cand = sg + step; sg = s_off[cand] <= (uint32_t)e ? cand : sg; }
      i = dl[(size_t)sg * (64 * kCertifyItems) + (e - (int)s_off[sg])];
    }
    bool hard = false, band = false;
    float4 srec = make_float4(0.f, 0.f, 0.f, 0.f);
    float drec = 0.f;
    int jrec = -1;
    listed_search_one(b, st, c, so, i, sub, L, s_hist, min_lb, hard, band_lo > 0, band_lo, band_hi, band, srec, drec, jrec);
    listed_append_hard(b, st, so, hard, i, lane);
    if (band_lo > 0) {                                     // the item's records: compacted in list order inside the workgroup
      const unsigned long long bm = __ballot(band);
      uint32_t* wc = s_wc[item & 1];
      if (lane == 0) wc[wave] = (uint32_t)__popcll(bm);
      __syncthreads();
      uint32_t before = 0, all = 0;
#pragma unroll
      for (int w2 = 0; w2 < kNnThreads / 64; ++w2) { const uint32_t n2 = wc[w2]; before += w2 < wave ? n2 : 0u; all += n2; }
      if (band) {
        const size_t at = (size_t)pair * 2 * b.bl_stride + (size_t)b.bl_stride + (size_t)chunk * qpb + before + __popcll(bm & ((1ull << lane) - 1ull));
        b.rec_a[at] = make_float4(srec.x, srec.y, srec.z, drec);
        b.rec_j[at] = jrec;
      }
      if (threadIdx.x == 0) b.gcount[(size_t)pair * b.seg_stride + nseg0 + chunk] = all;
    }
  }
  if (cur >= 0) flush();
}

// Fused iterations: the sums the certificate pass could not make, as ONE launch of a fixed grid of kSumsBlocks workgroups.
//  * A pair whose prediction held (spec_ok): the queries whose certificate failed this iteration (the per-wave segments of dlist the
//    fused pass left, dcount) have exact matches from the listed search now; those below the band are kept whatever the quantile
//    turns out to be inside it -- summed here, kListedSumChunk entries per work item, four per thread and round with their loads
//    issued level by level, into row nblk + item of partials.  (Lower-bounded ones lie above the quantile: nn_validate checked that,
//    or the iteration would not be in this form; an exact match with a coincident runner-up carries the bound 0 and is summed like
//    any other, as `accumulate` does.)  This was a phase of finalize -- one workgroup per pair walking 2 500 to 16 000 entries.
//  * A pair whose prediction missed: the plain `accumulate` of its every block (accumulate_block).  This was a standing launch of
//    `accumulate` per iteration, 3 840 workgroups of which all but a few pairs' returned at once.
// Workgroup id & 7 = XCD: it takes work of the pairs whose index in the launch has that residue, like xcd_block.
__device__ __forceinline__ void listed_sums_block(const IcpDev& b, PairState* st, int pair, int item, uint32_t* s_off, double (*s_red)[29], double* s_out) {
  const int ns = st->ns;
  const int nblk = (ns + kNnThreads * kCertifyItems - 1) / (kNnThreads * kCertifyItems);
  const int nseg0 = nblk * (kNnThreads / 64);
  const int32_t* dc = b.dcount + (size_t)pair * b.seg_stride;      // (listed_plan: entries before each segment, the total behind them)
  const int32_t* dl = b.dlist + (size_t)pair * b.dl_stride;
  int top = 1;
  while (2 * top < nseg0) top *= 2;
  for (int x = threadIdx.x; x < 2 * top; x += kAccThreads) s_off[x] = x < nseg0 ? (uint32_t)dc[x] : 0xffffffffu;
  const int nl = dc[nseg0];
  __syncthreads();
  const int band_lo = st->band_lo;
  const size_t so = (size_t)pair * b.ns_cap, to = (size_t)pair * b.nt_cap;
  double acc[29];
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = 0.0;
  // (a work item is long -- 16 entries per thread -- because it ends with a 29-column block reduction that costs as much as 8 entries
  // per thread; four entries per thread and round)
  constexpr int kL = 4;
  const int e_end = min(nl, (item + 1) * kListedSumChunk);
  for (int e0 = item * kListedSumChunk + (int)threadIdx.x; e0 < e_end; e0 += kL * kAccThreads) {
    int sg[kL], ii[kL], jj[kL];
    float dd[kL];
    float4 s4[kL], q4[kL], n4[kL];
#pragma unroll
    for (int k = 0; k < kL; ++k) sg[k] = 0;
    for (int step = top; step > 0; step >>= 1) {
#pragma unroll
      for (int k = 0; k < kL; ++k) {
        const int cand = sg[k] + step;
        sg[k] = s_off[cand] <= (uint32_t)min(e0 + kAccThreads * k, e_end - 1) ? cand : sg[k];
      }
    }
#pragma unroll
    for (int k = 0; k < kL; ++k) ii[k] = dl[(size_t)sg[k] * (64 * kCertifyItems) + (min(e0 + kAccThreads * k, e_end - 1) - (int)s_off[sg[k]])];
#pragma unroll
    for (int k = 0; k < kL; ++k) { s4[k] = ld_src(b, so + ii[k]); jj[k] = b.idx[so + ii[k]]; dd[k] = b.d2[so + ii[k]]; }
#pragma unroll
    for (int k = 0; k < kL; ++k) { q4[k] = b.tq[to + max(jj[k], 0)]; n4[k] = b.tn[to + max(jj[k], 0)]; }
#pragma unroll
    for (int k = 0; k < kL; ++k) {
      const uint32_t key = __float_as_uint(dd[k]);
      if (e0 + kAccThreads * k < e_end && jj[k] >= 0 && key < 0x7f800000u && (int)(key >> kHistShift) < band_lo)
        accumulate_terms(st->M, s4[k], q4[k], n4[k], acc);
    }
  }
  block_reduce29(acc, s_red, s_out);
  if (threadIdx.x < 29) b.partials[((size_t)pair * b.part_stride + nblk + item) * kAccCols + threadIdx.x] = s_out[threadIdx.x];
}

// ITEMS = points per thread of a missed pair's blocks (IcpDev::sums_items).  A block is a chain of dependent rounds and the launch
// lasts as long as its longest workgroup, so once only a few pairs miss (the band predictions hold from the third fused iteration
// on) the blocks are the short ones: an 8 192-point block takes ~40 us with the GPU nearly empty, a 2 048-point one ~12.  While most
// pairs still miss (no prediction yet, or a band of more than three bins) the long ones: a block's 29-column reduction and quantile
// look-up cost as much as 8 points per thread.
template <int ITEMS>
__global__ __launch_bounds__(kAccThreads) void iteration_sums(IcpDev b) {
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  __shared__ double s_red[4][29];
  __shared__ double s_out[29];
  __shared__ uint32_t s_plan[kListedMaxPairs / 8 + 1];       // work items before the k-th pair of this XCD
  __shared__ uint32_t s_off[2 * kFinalizeMaxSeg];
  const int x = (int)blockIdx.x & 7, lw = (int)blockIdx.x >> 3, nlw = (int)gridDim.x >> 3;
  const int npx = (b.npairs - x + 7) >> 3;                    // this XCD's pairs: x, x + 8, ... of the launch (<= kListedMaxPairs / 8)
  {
    uint32_t mine = 0;
    if ((int)threadIdx.x < npx) {
      const PairState* st = &b.state[b.pair_base + (int)threadIdx.x * 8 + x];
      if (!st->done) {
        if (st->spec_ok) mine = (st->deferred_count + (uint32_t)kListedSumChunk - 1u) / (uint32_t)kListedSumChunk;
        else mine = (uint32_t)((st->ns + kAccThreads * ITEMS - 1) / (kAccThreads * ITEMS));
      }
    }
    uint32_t total;
    const uint32_t o = block_excl_scan(mine, s_w, &total);
    if ((int)threadIdx.x < npx) s_plan[threadIdx.x] = o;
    if (threadIdx.x == 0) s_plan[npx] = total;
    __syncthreads();
  }
  const int nitems = (int)s_plan[npx];
  int lo = 0;
  for (int item = lw; item < nitems; item += nlw) {           // workgroup-uniform
    while (s_plan[lo + 1] <= (uint32_t)item) ++lo;
    const int pair = b.pair_base + lo * 8 + x;
    PairState* st = &b.state[pair];
    const int k = item - (int)s_plan[lo];
    if (st->spec_ok) {
      listed_sums_block(b, st, pair, k, s_off, s_red, s_out);
    } else {
      double Mc[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) Mc[c] = st->M[c];
      accumulate_block<ITEMS>(b, st, pair, k, Mc, s_w, s_q, s_red, s_out);
    }
    __syncthreads();
  }
}

// Reference-search mode, fused path: the queries the list walk (nn_nabo<., true>) has just matched again, by the fused pass's
// rule -- a match below the predicted band is summed, a match inside it becomes a record -- once the histogram is complete and
// nabo_validate has checked the prediction (the quantile's bin inside the band: this mode's nn_validate; spec_ok also tells
// `accumulate` to return at once, and finalize which form the iteration's sums have).  kNaboAccBlocks workgroups per pair take
// equal shares of the four class lists laid end to end; every wave leaves its records in its own segment of region 1 (a wave
// meets at most ceil(ns / 32) entries and a segment holds bl_stride / 32), the workgroup its sums in row nblk + blk of partials.
// (the check of the prediction: one workgroup per pair looks at the completed histogram -- a look costs as much as a small
// workgroup's whole share of the lists, so the kNaboAccBlocks workgroups of accumulate_listed read its verdict instead)
__global__ __launch_bounds__(kAccThreads) void nabo_validate(IcpDev b) {
  const int pair = b.pair_base + (int)blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  find_quantile_bin(b.hist + (size_t)pair * kHistBins, b.rho, s_w, s_q);
  if (threadIdx.x == 0) {
    const int band_lo = st->band_lo, band_hi = st->band_hi;
    st->spec_ok = (band_lo > 0 && s_q[2] > 0 && (int)s_q[0] >= band_lo && (int)s_q[0] <= band_hi) ? 1 : 0;
  }
}
__global__ __launch_bounds__(kAccThreads) void accumulate_listed(IcpDev b) {
  const int pair = b.pair_base + (int)blockIdx.y, blk = (int)blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
  __shared__ double s_red[4][29];
  __shared__ double s_out[29];
  const int ns = st->ns;
  const int nblk0 = (ns + kNnThreads * kCertifyItems - 1) / (kNnThreads * kCertifyItems);
  const int nseg0 = nblk0 * (kNnThreads / 64);
  int cls_first[5];
  cls_first[0] = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) cls_first[c + 1] = cls_first[c] + (int)st->nabo_count[c];
  const int total = cls_first[4];
  constexpr int kWaves = kNaboAccBlocks * (kAccThreads / 64);
  const int lane = threadIdx.x & 63, wave = blk * (kAccThreads / 64) + (int)(threadIdx.x >> 6);
  const int share = (total + kWaves - 1) / kWaves;
  const int e0 = wave * share, e1 = min(total, e0 + share);
  if (blk != 0 && blk * (kAccThreads / 64) * share >= total) {
    // nothing of the lists falls to this workgroup (the usual case once the pose has settled): empty segments, a zero row --
    // harmless if the prediction missed
    if (lane == 0) b.gcount[(size_t)pair * b.seg_stride + nseg0 + wave] = 0u;
    if (threadIdx.x < 29) b.partials[((size_t)pair * b.part_stride + nblk0 + blk) * kAccCols + threadIdx.x] = 0.0;
    return;
  }
  if (!st->spec_ok) return;                                 // (nabo_validate) no prediction, or a miss: `accumulate` sums the plain way
  const int band_lo = st->band_lo, band_hi = st->band_hi;
  const size_t so = (size_t)pair * b.ns_cap, to = (size_t)pair * b.nt_cap;
  const int seg_len1 = b.bl_stride / kWaves;
  const size_t segbase = (size_t)pair * 2 * b.bl_stride + (size_t)b.bl_stride + (size_t)wave * seg_len1;
  double Mc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  double acc[29];
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = 0.0;
  int wcount = 0;
  auto entry = [&](int e) -> int {                          // the e-th query of the four class lists laid end to end
    const int c = e < cls_first[1] ? 0 : (e < cls_first[2] ? 1 : (e < cls_first[3] ? 2 : 3));
    const int f = c == 0 ? cls_first[0] : (c == 1 ? cls_first[1] : (c == 2 ? cls_first[2] : cls_first[3]));
    return *nabo_list_slot(b, so, c, (uint32_t)(e - f));
  };
  constexpr int kU = 2;                                     // entries per lane and round: their dependent loads level by level
  for (int eb = e0; eb < e1; eb += 64 * kU) {               // wave-uniform
    int i[kU], jm[kU];
    float d[kU];
    float4 s4[kU], q4[kU], n4[kU];
    bool live[kU], below[kU], rec[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { live[u] = eb + 64 * u + lane < e1; i[u] = live[u] ? entry(eb + 64 * u + lane) : 0; }
#pragma unroll
    for (int u = 0; u < kU; ++u) { d[u] = b.d2[so + i[u]]; jm[u] = b.idx[so + i[u]]; s4[u] = ld_src(b, so + i[u]); }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t key = __float_as_uint(d[u]);
      const int bin = (int)(key >> kHistShift);
      const bool valid = live[u] && key < 0x7f800000u && jm[u] >= 0;
      below[u] = valid && bin < band_lo;
      rec[u] = valid && bin >= band_lo && bin <= band_hi;
      if (below[u]) { q4[u] = b.tq[to + jm[u]]; n4[u] = b.tn[to + jm[u]]; }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (below[u]) accumulate_terms(Mc, s4[u], q4[u], n4[u], acc);
      const unsigned long long bm = __ballot(rec[u]);
      if (rec[u]) {
        const size_t at = segbase + wcount + __popcll(bm & ((1ull << lane) - 1ull));
        b.rec_a[at] = make_float4(s4[u].x, s4[u].y, s4[u].z, d[u]);
        b.rec_j[at] = jm[u];
      }
      wcount += (int)__popcll(bm);
    }
  }
  if (lane == 0) b.gcount[(size_t)pair * b.seg_stride + nseg0 + wave] = (uint32_t)wcount;
  block_reduce29(acc, s_red, s_out);
  if (threadIdx.x < 29) b.partials[((size_t)pair * b.part_stride + nblk0 + blk) * kAccCols + threadIdx.x] = s_out[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// K4: exact quantile, solve, update, convergence
// ------------------------------------------------------------------------------------------
__device__ __noinline__ void jacobi_eig6(double* A, double* V, double* w) {
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) V[6 * i + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < 6; ++i) for (int j = i + 1; j < 6; ++j) off += A[6 * i + j] * A[6 * i + j];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        const double apq = A[6 * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[6 * q + q] - A[6 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) { const double akp = A[6 * k + p], akq = A[6 * k + q]; A[6 * k + p] = c * akp - s * akq; A[6 * k + q] = s * akp + c * akq; }
        for (int k = 0; k < 6; ++k) { const double apk = A[6 * p + k], aqk = A[6 * q + k]; A[6 * p + k] = c * apk - s * aqk; A[6 * q + k] = s * apk + c * aqk; }
        for (int k = 0; k < 6; ++k) { const double vkp = V[6 * k + p], vkq = V[6 * k + q]; V[6 * k + p] = c * vkp - s * vkq; V[6 * k + q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < 6; ++i) w[i] = A[6 * i + i];
}

// SolvePossiblyUnderdeterminedLinearSystem (icp_fast.cc:204-254): Cholesky when A is numerically
// invertible, otherwise the minimum-norm solution of the rank-reduced system (pseudo-inverse).
__device__ __forceinline__ void solve6(const double* A, const double* rhs, double* x) {
  // fully unrolled so that L lives in registers: with run-time indices it sits in scratch memory and every access is a
  // memory round trip (the serial tail of finalize took 9-19 us that way)
  double L[36];
  double dmax = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) dmax = fmax(dmax, fabs(A[6 * i + i]));
  bool ok = dmax > 0 && isfinite(dmax);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        ok = ok && (s > 1e-13 * dmax);
        L[6 * i + i] = sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  if (ok) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double s = rhs[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
      y[i] = s / L[6 * i + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
      x[i] = s / L[6 * i + i];
    }
    return;
  }
  double E[36], V[36], w[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) E[i] = A[i];
  jacobi_eig6(E, V, w);
  double wmax = 0;
  for (int i = 0; i < 6; ++i) wmax = fmax(wmax, fabs(w[i]));
  const double thresh = 2.220446049250313e-16 * 6 * wmax;
  for (int i = 0; i < 6; ++i) x[i] = 0;
  for (int k = 0; k < 6; ++k) {
    if (!(fabs(w[k]) > thresh)) continue;
    double c = 0;
    for (int i = 0; i < 6; ++i) c += V[6 * i + k] * rhs[i];
    c /= w[k];
    for (int i = 0; i < 6; ++i) x[i] += c * V[6 * i + k];
  }
}

__device__ void quat_from_rot(const double* T, double* q) {   // Eigen::Quaterniond(Matrix3d), T row-major 4x4
  const double r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[4], r11 = T[5], r12 = T[6], r20 = T[8], r21 = T[9], r22 = T[10];
  const double tr = r00 + r11 + r22;
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (r21 - r12) / s; q[2] = (r02 - r20) / s; q[3] = (r10 - r01) / s;
  } else if (r00 >= r11 && r00 >= r22) {
    const double s = sqrt(1.0 + r00 - r11 - r22) * 2;
    q[0] = (r21 - r12) / s; q[1] = 0.25 * s; q[2] = (r01 + r10) / s; q[3] = (r02 + r20) / s;
  } else if (r11 >= r22) {
    const double s = sqrt(1.0 + r11 - r00 - r22) * 2;
    q[0] = (r02 - r20) / s; q[1] = (r01 + r10) / s; q[2] = 0.25 * s; q[3] = (r12 + r21) / s;
  } else {
    const double s = sqrt(1.0 + r22 - r00 - r11) * 2;
    q[0] = (r10 - r01) / s; q[1] = (r02 + r20) / s; q[2] = (r12 + r21) / s; q[3] = 0.25 * s;
  }
}

__device__ double quat_angular_distance(const double* a, const double* c) {
  const double w = a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
  const double x = -a[0] * c[1] + a[1] * c[0] - a[2] * c[3] + a[3] * c[2];
  const double y = -a[0] * c[2] + a[2] * c[0] - a[3] * c[1] + a[1] * c[3];
  const double z = -a[0] * c[3] + a[3] * c[0] - a[1] * c[2] + a[2] * c[1];
  return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

// The serial end of an iteration, on ONE thread: the iteration's counters folded into the totals, the next search radius and
// quantile band, the 6x6 solve, the pose update with its motion potential, CheckConvergence and -- when the loop ends -- the result
// (icp_fast.cc:204-254, 306-323, 377-405, 506-528).  `st` is the state it reads and writes: the pair's in global memory for the
// `finalize` kernel; a workgroup's own LDS copy in the single-pair persistent kernel (icp_one.hip), where every workgroup runs
// this same tail on the same sums and only one of them (`publish`) touches what lies outside the state.  Returns the number of
// queries whose certificate failed in this iteration.
// (B: IcpDev, or TailOpts -- the handful of its fields the tail reads -- where the tail is a call, not inlined: icp_one.hip)
struct TailOpts {
  float cap_factor, ball_radius, band_gain, band_pad;
  int32_t fused, early_exit, max_iteration;
  uint32_t* search_hist;
  uint32_t* done_count;
};
template <class B>
__device__ __forceinline__ uint32_t finalize_tail(const B& b, PairState* st, int pair, const double* s_tot, uint32_t n_valid, uint32_t limit_key,
                                                  bool fusedm, int ns, bool publish) {
  st->fallback_total += st->unresolved_count;
  st->unresolved_count = 0;
  st->hard_total += st->hard_count;
  st->hard_count = 0;
  {
    const uint32_t walked = st->nabo_count[0] + st->nabo_count[1] + st->nabo_count[2] + st->nabo_count[3];   // SMHIP_NN_NABO's lists
    // (iteration 0 searches every query in every mode, whatever part of it went through a list)
    const uint32_t searched = st->iter == 0 ? (uint32_t)ns : (walked ? walked : (st->deferred_count ? st->deferred_count : (uint32_t)ns));
    st->searched_total += searched;
    if (publish && st->iter < kSearchHist) b.search_hist[(size_t)pair * kSearchHist + st->iter] = searched;
    for (int c = 0; c < 4; ++c) st->nabo_count[c] = 0;
  }
  const uint32_t n_listed = st->deferred_count;          // queries whose certificate failed in this iteration
  st->deferred_count = 0;
  st->min_lb_key = 0xffffffffu;
  st->refine = 0;
  {   // next search radius: cap_factor x the quantile distance, clamped to [0.05 m, ball_radius]
    const float lim = sqrtf(__uint_as_float(limit_key));
    float rc = fminf(fmaxf(b.cap_factor * lim, 0.05f), b.ball_radius);
    if (!(n_valid > 0)) rc = b.ball_radius;
    st->rcap2 = rc * rc;
  }
  {   // fused path: the band of histogram bins the NEXT iteration's quantile is expected in -- this quantile +- max(band_pad bins,
      // band_gain x its last move).  No previous quantile, a quantile in bin 0 or a band of more than three bins: no prediction
      // (the certificate pass then only certifies and `accumulate` sums).
    const uint32_t prev = st->limit_key;
    int lo = 0, hi = -1;
    if (n_valid > 0 && prev != 0u && limit_key != 0u) {
      const double x = (double)limit_key;
      const double w = fmax((double)b.band_gain * fabs(x - (double)prev), (double)b.band_pad * 1048576.0);
      const double l = x - w, u = x + w;
      if (l >= 1048576.0 && u < 2139095040.0) { lo = (int)((uint32_t)l >> kHistShift); hi = (int)((uint32_t)u >> kHistShift); }
      if (hi - lo > 2) { lo = 0; hi = -1; }
      // many certificates still fail (the pose still moves): the next iteration will list as many, which finalize would have to
      // walk -- no prediction, the certificate pass only certifies and `accumulate` sums
      if (b.fused && n_listed > (uint32_t)kFusedListedMax) { lo = 0; hi = -1; }
    }
    st->band_lo = lo; st->band_hi = hi;
    if (fusedm) st->spec_hits += 1;
    st->spec_ok = 0;
  }
  st->limit_key = limit_key;
  const double kept = s_tot[28];
  st->kept = (int)kept;
  if (n_valid == 0 || kept < 1.0) {       // icp_fast.cc:81 CHECK(!values.empty()) / :113 "no point to minimize"
    st->status = 6;                        // SMHIP_ERR_NO_MATCH
    st->done = 1;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) st->result[4 * c + r] = st->guess[4 * r + c];
    st->score = 0;
    if (publish) atomicAdd(b.done_count, 1u);
    return n_listed;
  }
  double A[36], rhs[6], x[6];
  {
    int c = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int e = a; e < 6; ++e) { A[6 * a + e] = s_tot[c]; A[6 * e + a] = s_tot[c]; ++c; }
#pragma unroll
    for (int a = 0; a < 6; ++a) rhs[a] = -s_tot[21 + a];                       // b = -(wF * dot), :302
  }
  solve6(A, rhs, x);                                                          // :304
  // transform = AngleAxis(|x[0:3]|, x[0:3] / |x[0:3]|), translation = x[3:6]    :306-312
  double dT[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  {
    const double ang = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const double ax = x[0] / ang, ay = x[1] / ang, az = x[2] / ang;
    const double c = cos(ang), s = sin(ang), v = 1.0 - c;
    double R[9] = {c + v * ax * ax, v * ax * ay - s * az, v * ax * az + s * ay,
                   v * ax * ay + s * az, c + v * ay * ay, v * ay * az - s * ax,
                   v * ax * az - s * ay, v * ay * az + s * ax, c + v * az * az};
    bool has_nan = false;
    for (int i = 0; i < 9; ++i) has_nan |= isnan(R[i]);
    for (int i = 0; i < 3; ++i) has_nan |= isnan(x[3 + i]);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) dT[4 * i + j] = has_nan ? ((i == j) ? 1.0 : 0.0) : R[3 * i + j];   // :315-321
    dT[3] = x[3]; dT[7] = x[4]; dT[11] = x[5];
  }
  double Tn[16];
  mat4_mul_rm(dT, st->T_iter, Tn);                                            // :506-510
  for (int i = 0; i < 16; ++i) st->T_iter[i] = Tn[i];
  double Mn[16];
  mat4_mul_rm(Tn, st->G, Mn);
  {   // motion potential: this step's ||dR|| and |dt| (what bounds |M_new s - M_old s| <= ||dR|| |s| + |dt|), and their running sums.
      // ||dR|| is the SPECTRAL norm of the 3x3 difference (the largest singular value: what |dR s| <= ||dR|| |s| needs), from the
      // closed-form largest eigenvalue of dR^T dR with 1e-6 of slack and never more than the Frobenius norm; for the difference of
      // two rotations the Frobenius norm is sqrt(2) times larger, and with it sqrt(2) times the certified motion of far points.
    double D[9], fa = 0, fb = 0;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { D[3 * r + c] = Mn[4 * r + c] - st->M[4 * r + c]; fa += D[3 * r + c] * D[3 * r + c]; }
      const double d = Mn[4 * r + 3] - st->M[4 * r + 3];
      fb += d * d;
    }
    double S[6];                                           // D^T D: xx xy xz yy yz zz
    S[0] = D[0] * D[0] + D[3] * D[3] + D[6] * D[6]; S[1] = D[0] * D[1] + D[3] * D[4] + D[6] * D[7]; S[2] = D[0] * D[2] + D[3] * D[5] + D[6] * D[8];
    S[3] = D[1] * D[1] + D[4] * D[4] + D[7] * D[7]; S[4] = D[1] * D[2] + D[4] * D[5] + D[7] * D[8]; S[5] = D[2] * D[2] + D[5] * D[5] + D[8] * D[8];
    double lam = fa;                                       // trace = squared Frobenius norm >= the largest eigenvalue
    {
      const double q = (S[0] + S[3] + S[5]) / 3.0;
      const double p1 = S[1] * S[1] + S[2] * S[2] + S[4] * S[4];
      const double p2 = (S[0] - q) * (S[0] - q) + (S[3] - q) * (S[3] - q) + (S[5] - q) * (S[5] - q) + 2.0 * p1;
      if (p2 > 0.0 && isfinite(p2)) {
        const double p = sqrt(p2 / 6.0), ip = 1.0 / p;
        const double b0 = (S[0] - q) * ip, b3 = (S[3] - q) * ip, b5 = (S[5] - q) * ip, b1 = S[1] * ip, b2 = S[2] * ip, b4 = S[4] * ip;
        double r = 0.5 * (b0 * (b3 * b5 - b4 * b4) - b1 * (b1 * b5 - b4 * b2) + b2 * (b1 * b4 - b3 * b2));
        r = fmin(1.0, fmax(-1.0, r));
        const double est = (q + 2.0 * p * cos(acos(r) / 3.0)) * (1.0 + 2e-6) + 1e-300;
        if (isfinite(est) && est > 0.0) lam = fmin(lam, est);
      } else if (p2 == 0.0) {
        lam = fmin(lam, q * (1.0 + 2e-6));
      }
    }
    st->step_a = sqrt(lam) * (1.0 + 1e-9); st->step_b = sqrt(fb) * (1.0 + 1e-9);
    st->pot_a += st->step_a; st->pot_b += st->step_b;
  }
  for (int i = 0; i < 12; ++i) { st->M_prev[i] = st->M[i]; st->M[i] = Mn[i]; }
  const int it = ++st->iter;                                                  // :513
  // history ring (latest at n_hist-1, at most 5 kept)
  int nh = st->n_hist;
  if (nh == 5) {
    for (int k = 0; k < 4; ++k) {
      for (int c = 0; c < 4; ++c) st->quat[k][c] = st->quat[k + 1][c];
      for (int c = 0; c < 3; ++c) st->trans[k][c] = st->trans[k + 1][c];
    }
    nh = 4;
  }
  quat_from_rot(Tn, st->quat[nh]);
  st->trans[nh][0] = Tn[3]; st->trans[nh][1] = Tn[7]; st->trans[nh][2] = Tn[11];
  st->n_hist = ++nh;
  bool converged = false;
  if (b.early_exit && nh > 4) {                                               // :377-405 (kSmoothLength = 4)
    double rd = 0, td = 0;
    for (int k = nh - 1; k >= nh - 4; --k) {
      rd += fabs(quat_angular_distance(st->quat[k], st->quat[k - 1]));
      const double dx = st->trans[k][0] - st->trans[k - 1][0], dy = st->trans[k][1] - st->trans[k - 1][1], dz = st->trans[k][2] - st->trans[k - 1][2];
      td += sqrt(dx * dx + dy * dy + dz * dz);
    }
    converged = (rd / 4 < 1e-3) && (td / 4 < 1e-2);
  }
  if (converged || it >= b.max_iteration) {                                   // :516-522
    // (the score of this iteration's kept matches, :518-521, is formed by final_score once the batch has left its loop)
    // result = T_mean * T_iter * T_mean^-1 * guess  (:527), written column-major
    double Tm[16] = {1, 0, 0, st->mu[0], 0, 1, 0, st->mu[1], 0, 0, 1, st->mu[2], 0, 0, 0, 1};
    double Rm[16];
    mat4_mul_rm(Tm, Mn, Rm);
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) st->result[4 * c + r] = Rm[4 * r + c];
    st->done = 1;
    if (publish) atomicAdd(b.done_count, 1u);
  }
  return n_listed;
}

__global__ __launch_bounds__(256) void finalize(IcpDev b) {
  const int pair = b.pair_base + blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
#if SMHIP_PHASE_TIMING
  if ((b.debug_flags & 16) && pair == 0 && threadIdx.x == 0) {            // phase timing report (diagnostic build only)
    unsigned long long* tc = reinterpret_cast<unsigned long long*>(b.tpart);
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += tc[k];
    printf("[phase] iter %d total %llu :", st->iter, tot);
    for (int k = 0; k < 8; ++k) { printf(" %.1f%%", tot ? 100.0 * (double)tc[k] / (double)tot : 0.0); tc[k] = 0; }
    printf("\n");
  }
#endif
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  __shared__ uint32_t s_h[256];
  __shared__ uint32_t s_sel[2];
  __shared__ double s_red[4][29];
  __shared__ double s_tot[29];
  __shared__ double s_part[8][32];
  __shared__ uint32_t s_keys[kFinalizeKeyCap];
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
#if SMHIP_PHASE_TIMING
  // diagnostic build, SMHIP_DEBUG_FLAGS=32: wall-clock stamps (100 MHz) of thread 0 of the launch's first pair at the phase boundaries
  const bool ftime = (b.debug_flags & 32) && threadIdx.x == 0;          // (prints the workgroups that took more than 60 us)
  unsigned long long fstamp[8];
  int fk = 0;
#define SMHIP_FPH() do { if (ftime && fk < 8) fstamp[fk++] = wall_clock64(); } while (0)
#else
#define SMHIP_FPH() do { } while (0)
#endif
  SMHIP_FPH();
  find_quantile_bin(gh, b.rho, s_w, s_q);
  SMHIP_FPH();
  const uint32_t qbin = s_q[0], below = s_q[1], n_valid = s_q[2], krank = s_q[3];
  const int ns = st->ns;
  // This iteration's sums come in one of two forms.  Plain: `accumulate` ran with the quantile's bin known -- rows [0, nblk) of
  // partials hold the sums below that bin, its waves' segments the bin's members.  Fused (b.fused and nn_validate's spec_ok):
  // the certificate pass summed its certified matches below a PREDICTED band of bins that does contain the quantile's -- rows
  // [0, nblk) -- and left records of the band's members; the listed search left records of the band's members among the
  // matches it found, and those of its matches that lie below the band were summed by iteration_sums (rows behind nblk).  Either
  // way a record carries all finalize needs (source point, d2, match), the records' order is fixed by the data alone, and
  // what is added from them is every record at or below the exact quantile: the band's lower bins entirely, the quantile's
  // bin up to the selected value.
  const bool fusedm = b.fused && st->spec_ok;
  // (a fused iteration's missed pairs were summed by iteration_sums in short blocks, whatever the batch's accumulate launches use)
  const int chunk = fusedm ? kNnThreads * kCertifyItems : kAccThreads * ((b.fused && !b.fused_nabo) ? b.sums_items : b.acc_items);
  const int nblk = (ns + chunk - 1) / chunk;
  const int nseg0 = nblk * 4;                                   // 4 waves per workgroup in both producers
  const int seg_len0 = chunk / 4;
  // (region 1: one segment per item of the listed search, as long as the item's queries -- listed_plan's cut of the list)
  // (reference-search mode: the kNaboAccBlocks x 4 waves of accumulate_listed, which also summed the listed matches below the band)
  const bool nabof = fusedm && b.fused_nabo;
  constexpr int kNaboSegs = kNaboAccBlocks * (kAccThreads / 64);
  const int nseg = nseg0 + (fusedm ? (nabof ? kNaboSegs : (int)b.litems[pair]) : 0);
  const int seg_len1 = nabof ? b.bl_stride / kNaboSegs : kNnThreads >> listed_lanes_log2((int)st->deferred_count, b.listed_lane_budget);
  // rows of partials behind the certificate pass's: reference-search mode: accumulate_listed's workgroups; else the listed matches
  // below the band, summed by iteration_sums kListedSumChunk entries per row (deferred_count = the length of the pair's list)
  const int nrows = nblk + (fusedm ? (nabof ? kNaboAccBlocks : (int)((st->deferred_count + (uint32_t)kListedSumChunk - 1u) / (uint32_t)kListedSumChunk)) : 0);
  // the rows of `partials` this thread folds at the end (every 8th from its group's first): loaded now, so that their latency runs
  // under the select phases instead of behind the last barrier (rows beyond the first 64 are read there)
  constexpr int kPrefRows = 8;
  double pref[kPrefRows];
  {
    const double* part = b.partials + (size_t)pair * b.part_stride * kAccCols + (threadIdx.x & 31);
#pragma unroll
    for (int u = 0; u < kPrefRows; ++u) {
      const int k = (int)(threadIdx.x >> 5) + 8 * u;
      pref[u] = k < nrows ? part[(size_t)k * kAccCols] : 0.0;
    }
  }
  const uint32_t* gcount = b.gcount + (size_t)pair * b.seg_stride;
  const float4* ra = b.rec_a + (size_t)pair * 2 * b.bl_stride;
  const int32_t* rj = b.rec_j + (size_t)pair * 2 * b.bl_stride;
  __shared__ uint32_t s_off[2 * kFinalizeMaxSeg];
  // where the k-th record of segment sg sits
  auto rec_at = [&](int sg, int k) -> uint32_t {
    return sg < nseg0 ? (uint32_t)(sg * seg_len0 + k) : (uint32_t)(b.bl_stride + (sg - nseg0) * seg_len1 + k);
  };
  auto segment_of = [&](int e) -> int {                   // the last segment with s_off[seg] <= e
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_off[mid] <= (uint32_t)e) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  auto member = [&](int e) -> uint32_t {
    const int sg = segment_of(e);
    return rec_at(sg, e - (int)s_off[sg]);
  };
  double acc[29];
#pragma unroll
  for (int c = 0; c < 29; ++c) acc[c] = 0.0;
  SMHIP_FPH();
  uint32_t limit_key = 0;
  int nb = 0;
  bool flat = false;
  int top = 1;                                             // the first step of the lock-step segment searches below
  if (n_valid > 0) {
    const int per = (nseg + 255) >> 8;                     // <= kFinalizeMaxSeg / 256
    const int s0 = (int)threadIdx.x * per;
    uint32_t c[kFinalizeMaxSeg / 256];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kFinalizeMaxSeg / 256; ++k) {
      c[k] = (k < per && s0 + k < nseg) ? gcount[s0 + k] : 0u;
      mine += c[k];
    }
    uint32_t total;
    uint32_t o = block_excl_scan(mine, s_w, &total);
#pragma unroll
    for (int k = 0; k < kFinalizeMaxSeg / 256; ++k)
      if (k < per && s0 + k < nseg) { s_off[s0 + k] = o; o += c[k]; }
    while (2 * top < nseg) top *= 2;
    for (int x = nseg + (int)threadIdx.x; x < 2 * top; x += 256) s_off[x] = 0xffffffffu;
    nb = (int)total;
    flat = nb <= kFinalizeKeyCap;
    __syncthreads();
    // lists that fit have their keys (d2's float bits) cached in LDS for the select passes; longer lists (very large clouds, or
    // many equal distances) are fetched again in every pass.  (Keys only: a record's position is found again by the same
    // LDS-only segment search when it is summed -- 8 192 keys instead of 4 096 key / position pairs in the same 32 KiB, and a
    // list that does not fit costs its workgroup ten times the others' time, which the whole launch then waits for.)
    if (flat && nb > 0) {
      constexpr int kPer = 16;                             // records per thread and round: two memory levels a round
      for (int r0 = 0; r0 < nb; r0 += kPer * 256) {        // workgroup-uniform
        int sg[kPer];
        uint32_t pos[kPer];
        float4 aa[kPer];
        // the kPer segment searches advance together, one LDS level per step (a search per entry, one after the other,
        // was most of this phase)
#pragma unroll
        for (int k = 0; k < kPer; ++k) sg[k] = 0;
        for (int step = top; step > 0; step >>= 1) {
#pragma unroll
          for (int k = 0; k < kPer; ++k) {
            // s_off is padded with 0xffffffff up to the next power of two: one unconditional read and a select per entry, so
            // the kPer reads of a step overlap (guarded by `cand < nseg` the compiler puts every read in its own branch)
            const int cand = sg[k] + step;
            sg[k] = s_off[cand] <= (uint32_t)min(r0 + (int)threadIdx.x + 256 * k, nb - 1) ? cand : sg[k];
          }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) pos[k] = rec_at(sg[k], min(r0 + (int)threadIdx.x + 256 * k, nb - 1) - (int)s_off[sg[k]]);
#pragma unroll
        for (int k = 0; k < kPer; ++k) aa[k] = ra[pos[k]];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          const int e = r0 + (int)threadIdx.x + 256 * k;
          if (e < nb) s_keys[e] = __float_as_uint(aa[k].w);
        }
      }
    }
    __syncthreads();
  }
  SMHIP_FPH();
  if (n_valid > 0) {
    // exact rank (krank - below) inside the quantile's bin: radix select on the low 20 key bits of that bin's records
    uint32_t rank = krank - below;
    uint32_t prefix = 0, mask = 0;
    const int shifts[3] = {12, 4, 0};
    const int widths[3] = {8, 8, 4};
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = shifts[pass];
      const uint32_t nd = 1u << widths[pass];
      s_h[threadIdx.x] = 0;
      __syncthreads();
      for (int e = threadIdx.x; e < nb; e += blockDim.x) {
        const uint32_t full = flat ? s_keys[e] : __float_as_uint(ra[member(e)].w);
        const uint32_t key = full & 0xfffffu;
        if ((full >> kHistShift) == qbin && (key & mask) == prefix) atomicAdd(&s_h[(key >> shift) & (nd - 1u)], 1u);
      }
      __syncthreads();
      {   // the digit whose cumulative count crosses `rank`, found by all 256 threads at once
        const uint32_t v = threadIdx.x < nd ? s_h[threadIdx.x] : 0u;
        uint32_t tot;
        const uint32_t excl = block_excl_scan(v, s_w, &tot);
        if (threadIdx.x == 0) { s_sel[0] = nd - 1; s_sel[1] = tot - s_h[nd - 1]; }
        __syncthreads();
        if (v > 0 && excl <= rank && rank < excl + v) { s_sel[0] = threadIdx.x; s_sel[1] = excl; }
      }
      __syncthreads();
      prefix |= s_sel[0] << shift;
      mask |= (nd - 1u) << shift;
      rank -= s_sel[1];
      __syncthreads();
    }
    limit_key = (qbin << kHistShift) | prefix;
    SMHIP_FPH();
    // weights = (d2 <= limit)  (icp_fast.cc:497-498): every record at or below the quantile.  (Keys order like the floats: a
    // record of a lower bin of the band is below the limit whatever its low bits.)
    const size_t to = (size_t)pair * b.nt_cap;
    if (flat) {
      // ten records per thread and round, their loads issued level by level (the record, then the matched target point +
      // normal) instead of one record after the other.  (Ten: one pair's quantile bin holds ~2 100 records, and with eight
      // the 30 left over cost a second round -- two more memory levels -- of the 28 us this kernel takes for one pair.)
      constexpr int kW = 10;
      for (int e0 = threadIdx.x; e0 < nb; e0 += kW * 256) {
        bool use[kW];
        uint32_t pos[kW];
        int jj[kW], sg[kW];
        float4 s4[kW], q4[kW], n4[kW];
#pragma unroll
        for (int k = 0; k < kW; ++k) {
          const int e = min(e0 + 256 * k, nb - 1);
          use[k] = e0 + 256 * k < nb && s_keys[e] <= limit_key;
          sg[k] = 0;
        }
        for (int step = top; step > 0; step >>= 1) {
#pragma unroll
          for (int k = 0; k < kW; ++k) {
            const int cand = sg[k] + step;
            sg[k] = s_off[cand] <= (uint32_t)min(e0 + 256 * k, nb - 1) ? cand : sg[k];
          }
        }
#pragma unroll
        for (int k = 0; k < kW; ++k) pos[k] = rec_at(sg[k], min(e0 + 256 * k, nb - 1) - (int)s_off[sg[k]]);
#pragma unroll
        for (int k = 0; k < kW; ++k) { s4[k] = ra[pos[k]]; jj[k] = max(rj[pos[k]], 0); }
#pragma unroll
        for (int k = 0; k < kW; ++k) { q4[k] = b.tq[to + jj[k]]; n4[k] = b.tn[to + jj[k]]; }
#pragma unroll
        for (int k = 0; k < kW; ++k)
          if (use[k]) accumulate_terms(st->M, s4[k], q4[k], n4[k], acc);
      }
    } else {
      for (int e = threadIdx.x; e < nb; e += blockDim.x) {
        const uint32_t at = member(e);
        const float4 a = ra[at];
        if (__float_as_uint(a.w) <= limit_key) {
          const int j = max(rj[at], 0);
          accumulate_terms(st->M, a, b.tq[to + j], b.tn[to + j], acc);
        }
      }
    }
  }
  SMHIP_FPH();
  block_reduce29(acc, s_red, s_tot);
  // add the per-block partial sums of the accumulate kernel: 8 thread groups take every 8th block,
  // then one thread per column folds the 8 group sums -- a fixed order, so the result is reproducible
  {
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double s = 0;
    const double* part = b.partials + (size_t)pair * b.part_stride * kAccCols + col;
#pragma unroll
    for (int u = 0; u < kPrefRows; ++u) if (grp + 8 * u < nrows) s += pref[u];       // (the same rows in the same order)
    for (int k = grp + 8 * kPrefRows; k < nrows; k += 8) s += part[(size_t)k * kAccCols];
    s_part[grp][col] = s;
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    double s = s_tot[threadIdx.x];
    for (int g = 0; g < 8; ++g) s += s_part[g][threadIdx.x];
    s_tot[threadIdx.x] = s;
  }
  // reset the per-iteration scratch for the next iteration
  for (int k = threadIdx.x; k < kHistBins; k += blockDim.x) gh[k] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  SMHIP_FPH();
  const uint32_t n_listed = finalize_tail(b, st, pair, s_tot, n_valid, limit_key, fusedm, ns, true);
  (void)n_listed;
#if SMHIP_PHASE_TIMING
  if (ftime && (wall_clock64() - fstamp[0]) > 6000ull) {
    const unsigned long long tend = wall_clock64();
    printf("[finalize] pair %d listed %u iter %d fused %d nb %d us: quantile %.1f listed %.1f records %.1f select %.1f sums %.1f fold %.1f tail %.1f total %.1f\n", pair, n_listed, st->iter, (int)fusedm, nb,
           (fstamp[1] - fstamp[0]) * 0.01, (fstamp[2] - fstamp[1]) * 0.01, (fstamp[3] - fstamp[2]) * 0.01, (fstamp[4] - fstamp[3]) * 0.01,
           (fstamp[5] - fstamp[4]) * 0.01, (fstamp[6] - fstamp[5]) * 0.01, (tend - fstamp[6]) * 0.01, (tend - fstamp[0]) * 0.01);
  }
#endif
}

// The score of the iteration an alignment left the loop with (icp_fast.cc:516-522: exp(-mean distance of that iteration's kept
// matches), formed there and nowhere else): one pass over the pair's distances of that iteration -- `d2` keeps them once the pair is
// done (every kernel returns on PairState::done) -- with the weights of :497-498, d2 <= limit (finalize's limit_key: keys order like
// the floats; a lower-bounded query's entry is its bound, which nn_validate keeps above the quantile).  kScoreParts workgroups per
// pair, each over a fixed range of the pair's points, leave their sums in PairState::score_part; score_fold (the next launch: no
// fences, no tickets) adds them in order.  The partition and every order of addition are fixed by ns alone.
__global__ __launch_bounds__(kAccThreads) void final_score(IcpDev b) {
  const int pair = b.pair_base + (int)(blockIdx.x / kScoreParts), part = (int)(blockIdx.x % kScoreParts);
  PairState* st = &b.state[pair];
  __shared__ double s_w4[kAccThreads / 64];
  __shared__ uint32_t s_c4[kAccThreads / 64];
  double s = 0.0;
  uint32_t cnt = 0;
  if (st->done && st->status == 0) {                      // (no match: finalize left score = 0, score_fold keeps it)
    const int ns = st->ns;
    const int len = ((ns + kScoreParts * kAccThreads - 1) / (kScoreParts * kAccThreads)) * kAccThreads;   // points per part: whole rounds
    const int lo = part * len, hi = min(ns, lo + len);
    const uint32_t limit_key = st->limit_key;
    const float* __restrict__ d2 = b.d2 + (size_t)pair * b.ns_cap;
    for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += 8 * kAccThreads) {
      float d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = d2[min(i0 + k * kAccThreads, hi - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * kAccThreads < hi && __float_as_uint(d[k]) <= limit_key) { s += sqrt((double)d[k]); ++cnt; }
    }
  }
  s = wave_sum_to_last(s);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 63) s_w4[threadIdx.x >> 6] = s;
  if ((threadIdx.x & 63) == 0) s_c4[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    st->score_part[part] = ((s_w4[0] + s_w4[1]) + s_w4[2]) + s_w4[3];
    st->score_cnt[part] = ((s_c4[0] + s_c4[1]) + s_c4[2]) + s_c4[3];
  }
}
__global__ void score_fold(IcpDev b, int npairs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npairs) return;
  PairState* st = &b.state[b.pair_base + k];
  if (!st->done || st->status != 0 || st->kept < 1) return;
  double tot = 0.0;
  uint32_t n = 0;
  for (int p = 0; p < kScoreParts; ++p) { tot += st->score_part[p]; n += st->score_cnt[p]; }
  // the mean over the matches that were summed: finalize's kept count is the same number whenever the distances of the last
  // iteration stand untouched (every iteration kernel returns on `done`); a difference is flagged, not divided away
  st->score_mismatch = n != (uint32_t)st->kept ? 1u : 0u;
  st->score = n ? exp(-tot / (double)n) : 0.0;
}

// Slot-to-slot copy of the uploaded clouds (benchmark replication).
__global__ void copy_slot(IcpDev b, float4* src, float4* tp, float4* tn, int from, int to) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.ns_cap) src[(size_t)to * b.ns_cap + i] = src[(size_t)from * b.ns_cap + i];
  if (i < b.nt_cap) {
    tp[(size_t)to * b.nt_cap + i] = tp[(size_t)from * b.nt_cap + i];
    tn[(size_t)to * b.nt_cap + i] = tn[(size_t)from * b.nt_cap + i];
  }
}

}  // namespace smhip
