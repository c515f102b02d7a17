// gicp_kernels.hip -- device side of the GICP stage of registrators::NdtWithGicp.
//
// /root/reference/registrators/ndt_gicp.cc:94-103 runs stock pcl::GeneralizedIterativeClosestPoint (PCL 1.8.1,
// not vendored).  The in-tree fork registrators/pclomp/gicp_omp_impl.hpp carries the same statements; lines below
// are the fork's:
//   gicp_knn_cov   computeCovariances                      :59-131
//   gicp_corr      correspondence filter + Mahalanobis     :441-463
//   gicp_fdf       OptimizationFunctorWithIndices f/df/fdf :250-377 (the sums; the 6-vector tail is host code)
// The BFGS minimiser (pcl/registration/bfgs.h) and the outer loop run on the host in smhip_gicp_api.hip.
#pragma once
#include "smhip_device.h"

namespace smhip {

constexpr int kGicpKMax = 32;                // k_correspondences_ <= 32 (default 20)
constexpr int kGicpKnnThreads = 128;
constexpr int kGicpCols = 16;                // f, g_t[3], R[9], count, 2 spare
constexpr int kGicpMaxBlocks = 1024;

constexpr int kGicpLaunchJobs = 16;          // jobs one gicp_corr / gicp_fdf launch carries in its arguments (a batch takes several launches)
constexpr int kGicpKnnJobs = 32;             // clouds one gicp_knn_cov launch covers

// A handle runs up to pair_slots / 2 NdtWithGicp JOBS side by side (smhip_ndt_gicp_align_batch; the single Align is job 0):
// job j works in pair slot j (down-sampled source and target, their search grid) and uses slot pair_slots / 2 + j as scratch
// for the source's own neighbour search.  Every array below holds one row per job.
struct GicpDev {
  double* cov_s;           // [jobs][ns_cap][6] source covariances, index = position in the job slot's src array
  double* cov_t;           // [jobs][nt_cap][6] target covariances, index = position in the job slot's tgt_p array
  double* maha;            // [jobs][ns_cap][6] mahalanobis_[i] (symmetric): xx xy xz yy yz zz
  float4* qraw;            // [jobs][ns_cap] matched raw target point; w = 1 if the correspondence is kept
  double* partials;        // [jobs][kGicpMaxBlocks][kGicpCols]
  double* out;             // [jobs][kGicpCols]
  uint32_t* ticket;        // [jobs] workgroups of the job's running gicp_fdf that have written their partial sums
  uint32_t* round_done;    // jobs of the running evaluation round whose sums are in host memory
  double* out_host;        // [jobs][kGicpCols] page-locked host memory: the folded sums; behind them ([jobs * kGicpCols]) the round's number
  uint32_t* count;         // [jobs] kept correspondences
  uint32_t* cov_epoch;     // [jobs][nt_cap] by the target point's position in the job slot's tgt_p array: the job's epoch when its covariance
                           //                was estimated (target covariances are estimated when first matched, not all up front)
  int32_t* need_list;      // [jobs][ns_cap] sorted positions of matched target points whose covariance is missing (this correspondence step)
  uint32_t* need_count;    // [jobs]
  int32_t jobs;            // rows
};

// symmetric 3x3 (xx xy xz yy yz zz) helpers
__device__ __forceinline__ void sym_inverse(const double* a, double* o) {
  const double c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id; o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
  o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}

// k nearest neighbours of every target point of `pair` among that same target (the point itself included, as
// nearestKSearch returns it), then the regularised covariance.  One query per thread; its best-k set lives in LDS
// (KMAX x 128 threads x 8 B) as an UNSORTED set plus, in registers, the largest distance in it and where that entry sits:
// a closer candidate overwrites that entry and one pass of k independent LDS reads finds the new largest.  (The sorted
// list this replaces shifted half the list through LDS on every accepted candidate -- dependent read-modify-write steps
// that cost several times the distance tests themselves: gicp_knn_cov was 56 % of NdtWithGicp::Align.)  The covariance is
// a sum over the set, so its order is irrelevant.  Search = shells of grid cells around the query's cell; inside a shell
// the x-extent of a grid row is one contiguous run of the cell-sorted target (row_slots).  A shell ends the
// search once the k-th distance is within the distance to the nearest unexplored face (block_guarantee).
// cov is indexed by the point's position in the pair's raw target array (tq.w), so it survives grid rebuilds.
// the slow path of gicp_knn_cov's neighbour set (ties at the set's largest distance; kept out of line so that the loop over
// the slots does not end up in the hot path's registers): among the members at distance `w`, the slot of the one with the
// largest caller index, and that index
template <int KMAX>
__device__ __noinline__ int knn_largest_index_at(const float (*s_d)[kGicpKnnThreads], const int (*s_j)[kGicpKnnThreads], int t, float w, int* slot) {
  int wo = -1, wp = 0;
  for (int m = 0; m < KMAX; ++m) if (s_d[m][t] == w) { const int o = s_j[m][t]; if (o > wo) { wo = o; wp = m; } }
  *slot = wp;
  return wo;
}

struct GicpKnnBatch {
  int32_t n, pad;
  int32_t slot[kGicpKnnJobs];            // pair slot whose TARGET is the cloud (a job's slot, or its scratch slot = the source)
  double* cov[kGicpKnnJobs];             // where its covariances go
};

// the neighbourhood and covariance of target point j0 (its sorted position) of pair slot `pair`; the workgroup's set columns
// s_d / s_j are the caller's
template <int KMAX>
__device__ __forceinline__ void gicp_knn_cov_one(const IcpDev& b, int pair, int j0, int k, double gicp_epsilon, double* cov,
                                                 float (*s_d)[kGicpKnnThreads], int (*s_j)[kGicpKnnThreads]) {
  const PairState* st = &b.state[pair];
  const int t = threadIdx.x;
  const float4* tq = b.tq + (size_t)pair * b.nt_cap;
  const uint2* words = b.words + (size_t)pair * kMaxGridWords;
  const uint32_t* cstart = b.cstart + (size_t)pair * (b.nt_cap + 1);
  const uint32_t* rowbits = b.have_rowbits ? b.rowbits + (size_t)pair * kMaxRowWords : nullptr;
  const float4 q = tq[j0];
#pragma unroll
  for (int m = 0; m < KMAX; ++m) { s_d[m][t] = m < k ? INFINITY : -2.0f; s_j[m][t] = -1; }   // entries >= k never hold the maximum
  float worst = INFINITY;                     // the largest distance of the set ...
  int wpos = 0;                               // ... and the entry that holds it
  const int cx = min(max(cell_coord(q.x, st->origin[0], st->inv_h), 0), st->nx - 1);
  const int cy = min(max(cell_coord(q.y, st->origin[1], st->inv_h), 0), st->ny - 1);
  const int cz = min(max(cell_coord(q.z, st->origin[2], st->inv_h), 0), st->nz - 1);
  auto scan = [&](int rowbase, int xa, int xb) {
    uint32_t sa, sb;
    row_slots(words, rowbase, xa, xb, sa, sb);
    if (sa == sb) return;                       // empty stretch of the row (most of a large shell): no run to fetch
    const uint32_t pa = cstart[sa], pb = cstart[sb];
    // The set is the k smallest by (distance, index in the caller's cloud).  Two candidates at the same float distance from the
    // query are not rare enough to ignore (a 37 k-point submap had one point whose 20th and 21st neighbours tie): with "first
    // visited stays" the set -- and that point's covariance, by 2e-2 -- followed the order of the points inside a grid cell,
    // which the builds of these grids leave to their atomics.  Ties only cost when they happen: a candidate AT the set's
    // largest distance, or an eviction that leaves another member at the evicted one's distance, takes the slow path.
    auto consider = [&](const float4 c, uint32_t) {
      const float d = dist2(c, q.x, q.y, q.z);
      if (!(d <= worst)) return;
      const int co = __float_as_int(c.w);
      if (d == worst) {
        if (!(worst < INFINITY)) return;
        int wp2;                                          // the member at this distance with the largest index ...
        const int wo = knn_largest_index_at<KMAX>(s_d, s_j, t, worst, &wp2);
        if (co < wo) s_j[wp2][t] = co;                    // ... makes way for a candidate with a smaller one (same distance: worst, wpos stand)
        return;
      }
      const float old = worst;
      const int old_o = s_j[wpos][t];
      s_d[wpos][t] = d; s_j[wpos][t] = co;              // evict the farthest member
      // new farthest member: KMAX independent LDS reads issued back to back (compile-time trip count: with a run-time
      // bound every read waited for the one before it, and a wave pays this whenever ANY of its lanes accepts a candidate)
      float dm[KMAX];
#pragma unroll
      for (int m = 0; m < KMAX; ++m) dm[m] = s_d[m][t];
      float w = -1.f;
      int wp = 0;
#pragma unroll
      for (int m = 0; m < KMAX; ++m) { const bool g = dm[m] > w; w = g ? dm[m] : w; wp = g ? m : wp; }
      if (w == old && old < INFINITY) {
        // another member sat at the evicted one's distance: of all of them the one with the largest index is the one to go
        int wp2;
        const int wo = knn_largest_index_at<KMAX>(s_d, s_j, t, w, &wp2);
        if (old_o < wo) s_j[wp2][t] = old_o;              // the evicted member comes back in that one's place
      }
      worst = w; wpos = wp;
    };
    uint32_t p = pa;
    for (; p + 4 <= pb; p += 4) {              // four candidate loads in flight
      const float4 c0 = tq[p], c1 = tq[p + 1], c2 = tq[p + 2], c3 = tq[p + 3];
      consider(c0, p); consider(c1, p + 1); consider(c2, p + 2); consider(c3, p + 3);
    }
    for (; p < pb; ++p) consider(tq[p], p);
  };
  // blocks of half-width 0, 1, 2, 4, ... cells around the query's cell; a larger block only visits what the previous one
  // did not (whole rows outside it, the two x extensions of the rows inside it), so sparse regions cost O(log) steps
  const int rmax = max(max(st->nx, st->ny), st->nz);
  const float row_slack = 2.0e-3f * st->h;       // (points sitting on a cell face, as in the 1-NN searches)
  int rp = -1;                                   // half-width already covered
  for (int r = 0; ; ) {
    const int X0 = max(cx - r, 0), X1 = min(cx + r, st->nx - 1);
    const int Y0 = max(cy - r, 0), Y1 = min(cy + r, st->ny - 1);
    const int Z0 = max(cz - r, 0), Z1 = min(cz + r, st->nz - 1);
    auto visit_row = [&](int z, int y) {
      // a row farther away than the set's largest distance holds no candidate (strictly farther: a point AT that distance may
      // still replace a member with a larger index); `worst` only shrinks, so what is skipped now stays irrelevant
      const float yl = st->origin[1] + (float)y * st->h, zl = st->origin[2] + (float)z * st->h;
      const float ry = fmaxf(fmaxf(yl - q.y, q.y - (yl + st->h)) - row_slack, 0.f);
      const float rz = fmaxf(fmaxf(zl - q.z, q.z - (zl + st->h)) - row_slack, 0.f);
      const float ryz2 = fmaf(ry, ry, rz * rz);
      if (ryz2 > worst) return;
      // ... and inside the row only the chord of that ball
      int xa = X0, xb = X1;
      if (worst < INFINITY) {
        const float rx = sqrtf(worst - ryz2) * 1.0001f + row_slack;
        xa = max(xa, cell_coord(q.x - rx, st->origin[0], st->inv_h));
        xb = min(xb, cell_coord(q.x + rx, st->origin[0], st->inv_h));
        if (xa > xb) return;
      }
      const int rowbase = (z * st->ny + y) * st->wx;
      const bool inner = rp >= 0 && abs(y - cy) <= rp && abs(z - cz) <= rp;
      if (!inner) {
        scan(rowbase, xa, xb);
      } else {
        const int xl1 = min(cx - rp - 1, xb), xr0 = max(cx + rp + 1, xa);
        if (xa <= xl1) scan(rowbase, xa, xl1);
        if (xr0 <= xb) scan(rowbase, xr0, xb);
      }
    };
    for (int z = Z0; z <= Z1; ++z) {
      // a shell of half-width r has (2 r + 1)^2 rows, nearly all empty around a far-range point: the row-occupancy bitmap
      // hands over the occupied ones (two words per slab and 32 rows) instead of two dependent loads per row
      if (rowbits) for_each_occupied_row(rowbits, st->ny, z, Y0, Y1, [&](int y) { visit_row(z, y); });
      else for (int y = Y0; y <= Y1; ++y) visit_row(z, y);
    }
    const float g = block_guarantee(st, q.x, q.y, q.z, X0, X1, Y0, Y1, Z0, Z1);
    if (g == INFINITY) break;                   // the block covers the grid
    if (g > 0.f && worst <= g * g) break;
    if (r > rmax) break;
    rp = r;
    // the next block: once the set is full its largest distance bounds the neighbourhood, and the block that reaches that far
    // ends the search -- doubling past it (a far-range point with metres to its 20th neighbour: half-width 8 -> 16 -> 32) visits
    // up to eight times the rows for nothing
    const int dbl = r == 0 ? 1 : 2 * r;
    r = worst < INFINITY ? max(r + 1, min(dbl, (int)(sqrtf(worst) * st->inv_h) + 2)) : dbl;
  }
  // covariance of the k neighbours from the RAW coordinates; the products pt.x * pt.y are float products (:95-103)
  // The set's slots are filled in visiting order, which follows the order of the points inside a grid cell: summed slot by
  // slot the covariance would differ in its last bits between two builds of the same grid.  The neighbours (s_j holds their
  // indices in the caller's cloud) are summed in index order instead: k passes over the k slots.
  const float4* raw = b.tgt_p + (size_t)pair * b.nt_cap;
  double mean[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  int last = -1;
  int kv = 0;                                   // members actually summed
  for (int m = 0; m < k; ++m) {
    int nxt = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < KMAX; ++e) { const int o = s_j[e][t]; nxt = (e < k && o > last && o < nxt) ? o : nxt; }
    // fewer than k members: the cloud holds fewer than k FINITE points (a non-finite candidate's distance is NaN and never
    // enters the set; the host only checks the raw counts), or the query itself is not finite.  No member left to fetch.
    if (nxt == 0x7fffffff) break;
    last = nxt;
    ++kv;
    const float4 p = raw[nxt];
    mean[0] += p.x; mean[1] += p.y; mean[2] += p.z;
    const float xx = p.x * p.x, yx = p.y * p.x, yy = p.y * p.y, zx = p.z * p.x, zy = p.z * p.y, zz = p.z * p.z;
    c[0] += (double)xx; c[1] += (double)yx; c[2] += (double)zx; c[3] += (double)yy; c[4] += (double)zy; c[5] += (double)zz;
  }
  double* o = cov + (size_t)__float_as_int(q.w) * 6;
  if (kv == 0) {                                // nothing to estimate a surface from: the isotropic covariance
    o[0] = 1.0; o[1] = 0.0; o[2] = 0.0; o[3] = 1.0; o[4] = 0.0; o[5] = 1.0;
    return;
  }
  const double kk = (double)kv;                 // = k whenever the neighbourhood is complete (every finite cloud of >= k points)
  for (int a = 0; a < 3; ++a) mean[a] /= kk;
  double A[9], V[9], w[3];
  A[0] = c[0] / kk - mean[0] * mean[0];
  A[1] = A[3] = c[1] / kk - mean[1] * mean[0];
  A[2] = A[6] = c[2] / kk - mean[2] * mean[0];
  A[4] = c[3] / kk - mean[1] * mean[1];
  A[5] = A[7] = c[4] / kk - mean[2] * mean[1];
  A[8] = c[5] / kk - mean[2] * mean[2];
  jacobi_eig3(A, V, w);                         // ascending eigenvalues, eigenvectors in the columns of V
  // JacobiSVD orders by singular value = |eigenvalue|: the direction that gets gicp_epsilon is the smallest |w|
  int col = 0;
  if (fabs(w[1]) < fabs(w[col])) col = 1;
  if (fabs(w[2]) < fabs(w[col])) col = 2;
  const double u0 = V[col], u1 = V[3 + col], u2 = V[6 + col];
  const double s = 1.0 - gicp_epsilon;          // U diag(1, 1, eps) U^T = I - (1 - eps) u3 u3^T  (:118-129)
  o[0] = 1.0 - s * u0 * u0; o[1] = -s * u0 * u1; o[2] = -s * u0 * u2;
  o[3] = 1.0 - s * u1 * u1; o[4] = -s * u1 * u2; o[5] = 1.0 - s * u2 * u2;
}

// every point of the clouds of L (a source through its scratch slot; a whole target: the parity hook)
template <int KMAX>
__global__ __launch_bounds__(kGicpKnnThreads) void gicp_knn_cov(IcpDev b, GicpKnnBatch L, int k, double gicp_epsilon) {
  __shared__ float s_d[KMAX][kGicpKnnThreads];
  __shared__ int s_j[KMAX][kGicpKnnThreads];
  const int pair = L.slot[blockIdx.y];
  const int j0 = blockIdx.x * kGicpKnnThreads + threadIdx.x;
  if (j0 >= b.state[pair].nt) return;
  gicp_knn_cov_one<KMAX>(b, pair, j0, k, gicp_epsilon, L.cov[blockIdx.y], s_d, s_j);
}

// Target covariances on demand.  A GICP run consumes the covariance of a target point only when a source point is matched to it
// (:449-459) -- some 34 000 of the 574 000 points of the config #5 submap -- so they are estimated when first needed instead of
// all up front (computeCovariances for the whole target, :391-402: 0.65-1.25 ms per Align): gicp_need lists the matched target
// points of a correspondence step whose covariance does not carry the job's current epoch (and stamps them), gicp_knn_cov_listed
// estimates those -- the same function of the same neighbours, so every value that is used is the value the full pass gives.
struct GicpNeedJob { int32_t job, ns; float thr2; uint32_t epoch; };
struct GicpNeedBatch {
  int32_t n, k;
  double gicp_epsilon;
  GicpNeedJob j[kGicpLaunchJobs];
};
__global__ __launch_bounds__(256) void gicp_need(IcpDev b, GicpDev g, GicpNeedBatch L) {
  const GicpNeedJob& J = L.j[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t so = (size_t)J.job * b.ns_cap, to = (size_t)J.job * b.nt_cap;
  bool need = false;
  int j = -1;
  if (i < J.ns) {
    j = b.idx[so + i];
    if (j >= 0 && b.d2[so + i] < J.thr2) {
      const int orig = __float_as_int(b.tq[to + j].w);
      need = atomicExch(&g.cov_epoch[to + orig], J.epoch) != J.epoch;      // the first to ask lists the point
    }
  }
  const unsigned long long m = __ballot(need);
  if (m) {
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(g.need_count + J.job, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    if (need) g.need_list[so + base + __popcll(m & ((1ull << lane) - 1ull))] = j;
  }
}
template <int KMAX>
__global__ __launch_bounds__(kGicpKnnThreads) void gicp_knn_cov_listed(IcpDev b, GicpDev g, GicpNeedBatch L) {
  __shared__ float s_d[KMAX][kGicpKnnThreads];
  __shared__ int s_j[KMAX][kGicpKnnThreads];
  const GicpNeedJob& J = L.j[blockIdx.y];
  const uint32_t e = blockIdx.x * kGicpKnnThreads + threadIdx.x;
  if (e >= g.need_count[J.job]) return;
  const int j0 = g.need_list[(size_t)J.job * b.ns_cap + e];
  gicp_knn_cov_one<KMAX>(b, J.job, j0, L.k, L.gicp_epsilon, g.cov_t + (size_t)J.job * b.nt_cap * 6, s_d, s_j);
}

// per source point: keep the correspondence if d2 < threshold^2 and store (R C1 R^T + C2)^-1 and the raw target point
struct GicpCorrJob {
  int32_t job, ns;
  float thr2, pad;
  double R[9];             // rotation of transformation_ * guess, row-major (:425-429)
};
struct GicpCorrBatch {
  int32_t n, pad;
  GicpCorrJob j[kGicpLaunchJobs];
};
__global__ __launch_bounds__(256) void gicp_corr(IcpDev b, GicpDev g, GicpCorrBatch L) {
  const GicpCorrJob& J = L.j[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t so = (size_t)J.job * b.ns_cap, to = (size_t)J.job * b.nt_cap;
  bool keep = false;
  if (i < J.ns) {
    const int j = b.idx[so + i];
    const float d2 = b.d2[so + i];
    if (j >= 0 && d2 < J.thr2) {                                                     // :449
      keep = true;
      const int orig = __float_as_int(b.tq[to + j].w);
      const double* C1 = g.cov_s + (so + i) * 6;
      const double* C2 = g.cov_t + (to + orig) * 6;
      const double* Rrm = J.R;
      const double c1[9] = {C1[0], C1[1], C1[2], C1[1], C1[3], C1[4], C1[2], C1[4], C1[5]};
      double M[9];                                                                   // M = R * C1
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[3 * r + c] = Rrm[3 * r] * c1[c] + Rrm[3 * r + 1] * c1[3 + c] + Rrm[3 * r + 2] * c1[6 + c];
      double T[6];                                                                   // temp = M * R^T + C2 (symmetric)
      int e = 0;
      for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
          const double full[9] = {C2[0], C2[1], C2[2], C2[1], C2[3], C2[4], C2[2], C2[4], C2[5]};
          T[e++] = M[3 * r] * Rrm[3 * c] + M[3 * r + 1] * Rrm[3 * c + 1] + M[3 * r + 2] * Rrm[3 * c + 2] + full[3 * r + c];
        }
      sym_inverse(T, g.maha + (so + i) * 6);                                          // :459
      float4 p = b.tgt_p[to + orig];
      p.w = 1.f;
      g.qraw[so + i] = p;
    } else {
      g.qraw[so + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(g.count + J.job, (uint32_t)__popcll(m));
}

struct GicpPose {
  float T[12];             // applyState(base_transformation_, x), rows 0..2
  float B[12];             // base_transformation_ (the guess), rows 0..2
};
struct GicpFdfJob {
  int32_t job, ns, nblk, pad;          // nblk = min(kGicpMaxBlocks, ceil(ns / 256)): the job's own workgroups, whatever the launch holds
  GicpPose P;
};
struct GicpFdfBatch {
  int32_t n, total;                    // jobs in this launch / in the whole round (a round of more than kGicpLaunchJobs takes several)
  unsigned long long seq;              // the round's number
  GicpFdfJob j[kGicpLaunchJobs];
};

// sums of the functor over the kept correspondences: f (:272), g_t (:316-318), R (:320-321), for every job of the round
// (blockIdx.y) at that job's own pose.  A job's sums are formed by its own nblk workgroups in its own order, so they do not
// depend on which other jobs share the launch.  The workgroup of a job that finishes last folds the job's partial sums
// (16 strided groups, then the groups in turn) and stores them straight into page-locked host memory; the job that finishes last stores the round's
// number after them: the host, which needs f and g before it can choose any job's next point, spins on that number instead
// of paying for a second launch, a copy and a stream synchronise per evaluation.
__global__ __launch_bounds__(256) void gicp_fdf(IcpDev b, GicpDev g, GicpFdfBatch L) {
  const GicpFdfJob& J = L.j[blockIdx.y];
  const int nblk = J.nblk, ns = J.ns, job = J.job;
  if ((int)blockIdx.x >= nblk) return;
  const GicpPose& P = J.P;
  const size_t so = (size_t)job * b.ns_cap;
  double acc[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) acc[k] = 0.0;
  double cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += nblk * blockDim.x) {
    const float4 q = g.qraw[so + i];
    if (q.w == 0.f) continue;
    const float4 s = b.src[so + i];
    // Eigen Matrix4f * Vector4f: column by column accumulation in float
    const float px = ((P.T[0] * s.x + P.T[1] * s.y) + P.T[2] * s.z) + P.T[3];
    const float py = ((P.T[4] * s.x + P.T[5] * s.y) + P.T[6] * s.z) + P.T[7];
    const float pz = ((P.T[8] * s.x + P.T[9] * s.y) + P.T[10] * s.z) + P.T[11];
    const double r0 = (double)(px - q.x), r1 = (double)(py - q.y), r2 = (double)(pz - q.z);      // float differences (:268)
    const double* M = g.maha + (so + i) * 6;
    const double t0 = M[0] * r0 + M[1] * r1 + M[2] * r2;
    const double t1 = M[1] * r0 + M[3] * r1 + M[4] * r2;
    const double t2 = M[2] * r0 + M[4] * r1 + M[5] * r2;
    acc[0] += r0 * t0 + r1 * t1 + r2 * t2;
    acc[1] += t0; acc[2] += t1; acc[3] += t2;
    const double bx = (double)(((P.B[0] * s.x + P.B[1] * s.y) + P.B[2] * s.z) + P.B[3]);         // base_transformation_ * p_src
    const double by = (double)(((P.B[4] * s.x + P.B[5] * s.y) + P.B[6] * s.z) + P.B[7]);
    const double bz = (double)(((P.B[8] * s.x + P.B[9] * s.y) + P.B[10] * s.z) + P.B[11]);
    acc[4] += bx * t0; acc[5] += bx * t1; acc[6] += bx * t2;                                      // R += p_src3 * temp^T
    acc[7] += by * t0; acc[8] += by * t1; acc[9] += by * t2;
    acc[10] += bz * t0; acc[11] += bz * t1; acc[12] += bz * t2;
    cnt += 1.0;
  }
  __shared__ double s_red[4][kGicpCols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 13; ++k) acc[k] = wave_sum(acc[k]);
  cnt = wave_sum(cnt);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 13; ++k) s_red[wave][k] = acc[k];
    s_red[wave][13] = cnt; s_red[wave][14] = 0; s_red[wave][15] = 0;
  }
  __syncthreads();
  double* partials = g.partials + (size_t)job * kGicpMaxBlocks * kGicpCols;
  if (threadIdx.x < kGicpCols) {
    partials[(size_t)blockIdx.x * kGicpCols + threadIdx.x] =
        s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    __threadfence();
  }
  __shared__ uint32_t s_last;
  __shared__ double s_g[16][kGicpCols];
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(g.ticket + job, 1u) == (uint32_t)nblk - 1u ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  {
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;            // 16 strided groups, then the groups in turn
    double t = 0;
    for (int k = grp; k < nblk; k += 16) t += partials[(size_t)k * kGicpCols + c];
    s_g[grp][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < kGicpCols) {
    double t = 0;
    for (int k = 0; k < 16; ++k) t += s_g[k][threadIdx.x];
    g.out[(size_t)job * kGicpCols + threadIdx.x] = t;
    g.out_host[(size_t)job * kGicpCols + threadIdx.x] = t;
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    g.ticket[job] = 0;
    __threadfence_system();
    if (atomicAdd(g.round_done, 1u) == (uint32_t)L.total - 1u) {        // every job of the round has its sums in host memory
      *g.round_done = 0;
      __threadfence_system();
      *reinterpret_cast<volatile unsigned long long*>(g.out_host + (size_t)g.jobs * kGicpCols) = L.seq;
    }
  }
}

__global__ void gicp_copy_points(const float4* from, float4* to, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) to[i] = from[i];
}

}  // namespace smhip
