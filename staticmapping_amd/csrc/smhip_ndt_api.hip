// smhip_ndt_api.hip -- host side of registrators::Ndt on the C ABI (included by smhip_api.hip).
//
// Ndt::Align (/root/reference/registrators/ndt.cc:38-64) = convert clouds, setInputTarget (voxel grid
// build, every call), pclomp NDT align, getFitnessScore.  The 6-vector Newton / More-Thuente driver
// (pclomp/ndt_omp_impl.hpp:81-171, 757-916) runs here on the host exactly as in the reference, as a
// state machine per pair: K Aligns (smhip_ndt_align_batch; smhip_ndt_align is the batch of one) advance in
// lock-step, and every round's computeDerivatives calls -- one per pair still running, each at the pose ITS
// line search asks for -- are ONE ndt_derivatives + ndt_reduce launch (grid.y = pair) and one read-back.  Every
// pair's evaluation sequence is exactly the one the reference walks; only the launches are shared.
//
// Provenance note: computeStepLengthMT / trialValueSelectionMT / updateIntervalMT below are a PORT, not a redesign -- ~80
// lines of scalar host control flow that follow pclomp/ndt_omp_impl.hpp:633-916 branch for branch (same variable roles:
// a_l, f_l, g_l, a_u, phi_0, d_psi_t, open_interval ...), including the reference's quirk that `interval_converged` is
// computed from the UN-updated interval.  The iteration and derivative-call counts of the parity tests depend on that control
// flow being reproduced exactly; nothing in it is data-parallel.  Everything it calls (the derivative evaluations, the voxel
// table, the fitness search) is this repository's own GPU code.
#include "ndt_kernels.hip"

namespace {

struct NdtSlotMeta {                  // what the voxel table resident in a slot was built from
  bool valid = false;
  unsigned long long gen = 0;         // tgt_gen[slot]
  float resolution = 0.f; int min_points = 0; float eig_mult = 0.f;
};

struct NdtHost {
  int cap = 0;                        // pair slots [0, cap) have table storage
  NdtDev* devs_dev = nullptr;         // [cap] device array the kernels index
  NdtDev* devs_host = nullptr;        // [cap] page-locked mirror
  NdtGridInfo* info_all = nullptr;    // per-slot storage, contiguous: [cap] ...
  uint32_t* bits_all = nullptr;       // [cap][kNdtMaxWords]
  uint2* words_all = nullptr;
  uint32_t* vstart_all = nullptr;     // [cap][nt_cap + 1]
  float4* vpts_all = nullptr;         // [cap][nt_cap]
  NdtVoxel* vox_all = nullptr;
  double* icovd_all = nullptr;        // [cap][nt_cap][6]
  double* partials_all = nullptr;     // [cap][kNdtMaxDerivBlocks][kNdtDerivCols]
  double* out_all = nullptr;          // [cap][kNdtDerivCols]
  NdtPose* poses_dev = nullptr; NdtPose* poses_host = nullptr;        // [cap] this round's evaluations
  int32_t* active_dev = nullptr; int32_t* active_host = nullptr;      // [cap] the table each evaluation runs against
  int32_t* ns_dev = nullptr;                                           // [cap] source sizes (fitness pass)
  double* out_pinned = nullptr;       // [cap][kNdtDerivCols]
  NdtGridInfo* info_pinned = nullptr; // [cap]
  double* fit_dev = nullptr;          // [cap][128]
  double* fit_pinned = nullptr;
  int32_t* vkey = nullptr;            // [nt_cap] linear voxel index of every occupied voxel of slot 0 (tests)
  PrepWorkspace* prep = nullptr;      // radix-sort workspace for cap * nt_cap (voxel code, point) pairs
  std::vector<void*> dev_allocs, host_allocs;
  std::vector<NdtSlotMeta> meta;
  smhip_ndt_options opts{};
  int deriv_calls = 0;                // of the last single Align (statistics)
  double last_pairs = 0;
  bool double_math = false;           // stock pcl::NormalDistributionsTransform arithmetic (NdtWithGicp)
};

// ---- small dense helpers (host, double unless noted) ------------------------------------------
void pose_to_matrix_f32(const double* p, float* T /*row-major 4x4*/) {
  // Translation(p0..2) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z), all float (:146-149, :803-806)
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  const float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  const float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float M[9], R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += Rx[3 * i + k] * Ry[3 * k + j]; M[3 * i + j] = s; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += M[3 * i + k] * Rz[3 * k + j]; R[3 * i + j] = s; }
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = (float)p[i]; }
  T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

void euler_xyz_f32(const float* T /*row-major 4x4*/, float* e) {
  // Eigen 3.3 eulerAngles(0, 1, 2) on a Matrix3f (:109)
  auto R = [&](int r, int c) { return T[4 * r + c]; };
  float r0 = std::atan2(R(1, 2), R(2, 2));
  const float c2 = std::sqrt(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
  float r1;
  if (r0 > 0.f) { r0 -= (float)M_PI; r1 = std::atan2(-R(0, 2), -c2); }
  else r1 = std::atan2(-R(0, 2), c2);
  const float s1 = std::sin(r0), c1 = std::cos(r0);
  const float r2 = std::atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
  e[0] = -r0; e[1] = -r1; e[2] = -r2;
}

// Eigen::JacobiSVD<Matrix6d>(H, FullU | FullV).solve(b): one-sided Jacobi SVD + pseudo-inverse (:127-129)
void svd_solve6(const double* H /*row-major*/, const double* b, double* x) {
  double A[36], V[36];
  for (int i = 0; i < 36; ++i) { A[i] = H[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int k = 0; k < 6; ++k) { app += A[6 * k + p] * A[6 * k + p]; aqq += A[6 * k + q] * A[6 * k + q]; apq += A[6 * k + p] * A[6 * k + q]; }
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq)) continue;
        rotated = true;
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
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
  for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += A[6 * k + j] * A[6 * k + j]; sig[j] = std::sqrt(s); smax = std::max(smax, sig[j]); }
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

void angle_derivatives(const double* p, NdtPose& P) {   // :288-393
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
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
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) { P.j_ang[r][c] = (float)j[r][c]; P.j_angd[r][c] = j[r][c]; }
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) { P.h_ang[r][c] = (float)hh[r][c]; P.h_angd[r][c] = hh[r][c]; }
}

}  // namespace

struct smhip_ndt_state { NdtHost n; };

namespace {

NdtHost& ndt_of(smhip_context* h);

void ndt_release(NdtHost& n) {
  for (void* p : n.dev_allocs) (void)hipFree(p);
  for (void* p : n.host_allocs) (void)hipHostFree(p);
  n.dev_allocs.clear(); n.host_allocs.clear();
  if (n.prep) { prep_destroy(n.prep); n.prep = nullptr; }
  n.cap = 0; n.meta.clear();
}

// table storage for pair slots [0, need): allocated on first use, re-allocated (all tables dropped) when a later batch needs more
smhip_status ndt_ensure(smhip_context* h, int need = 1) {
  NdtHost& n = ndt_of(h);
  if (need <= n.cap) return SMHIP_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ndt_release(n);
  const size_t K = (size_t)need, NT = h->dev.nt_cap;
  bool ok = true;
  auto D = [&](auto** p, size_t count) {
    void* v = nullptr;
    if (ok && hipMalloc(&v, count * sizeof(**p)) == hipSuccess) { n.dev_allocs.push_back(v); *p = reinterpret_cast<decltype(*p)>(v); } else ok = false;
  };
  auto P = [&](auto** p, size_t count) {
    void* v = nullptr;
    if (ok && hipHostMalloc(&v, count * sizeof(**p)) == hipSuccess) { n.host_allocs.push_back(v); *p = reinterpret_cast<decltype(*p)>(v); } else ok = false;
  };
  D(&n.devs_dev, K); D(&n.info_all, K); D(&n.bits_all, K * kNdtMaxWords); D(&n.words_all, K * kNdtMaxWords);
  D(&n.vstart_all, K * (NT + 1)); D(&n.vpts_all, K * NT); D(&n.vox_all, K * NT); D(&n.icovd_all, K * NT * 6);
  D(&n.partials_all, K * (size_t)kNdtMaxDerivBlocks * kNdtDerivCols); D(&n.out_all, K * kNdtDerivCols);
  D(&n.poses_dev, K); D(&n.active_dev, K); D(&n.ns_dev, K); D(&n.fit_dev, K * 128); D(&n.vkey, NT);
  P(&n.devs_host, K); P(&n.poses_host, K); P(&n.active_host, K); P(&n.out_pinned, K * kNdtDerivCols); P(&n.info_pinned, K); P(&n.fit_pinned, K * 128);
  if (ok && K * NT <= (size_t)0x7fffffff) { n.prep = prep_create((int)(K * NT)); ok = n.prep != nullptr; } else ok = false;
  if (!ok) { ndt_release(n); h->err = "NDT table allocation failed"; return SMHIP_ERR_HIP; }
  n.cap = need;
  n.meta.assign(K, NdtSlotMeta{});
  for (size_t k = 0; k < K; ++k) {
    NdtDev& d = n.devs_host[k];
    d = NdtDev{};
    d.info = n.info_all + k; d.bits = n.bits_all + k * kNdtMaxWords; d.words = n.words_all + k * kNdtMaxWords;
    d.vstart = n.vstart_all + k * (NT + 1); d.vpts = n.vpts_all + k * NT; d.vox = n.vox_all + k * NT; d.icovd = n.icovd_all + k * NT * 6;
    d.partials = n.partials_all + k * (size_t)kNdtMaxDerivBlocks * kNdtDerivCols; d.out = n.out_all + k * kNdtDerivCols;
    d.tgt = h->dev.tgt_p + k * NT; d.src = h->dev.src + k * (size_t)h->dev.ns_cap; d.tpart = h->dev.tpart + k * kTgtReduceBlocks * 16;
  }
  return SMHIP_OK;
}

__global__ void ndt_voxel_keys(const NdtDev* devs, int32_t* keys) {
  const NdtDev d = devs[0];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.info->nocc) return;
  const float4 p = d.vpts[d.vstart[v]];
  int i0, i1, i2;
  ndt_voxel_of(d.info, p.x, p.y, p.z, i0, i1, i2);
  keys[v] = i0 + i1 * d.info->div_b[0] + i2 * d.info->div_b[0] * d.info->div_b[1];   // :223
}

bool ndt_table_current(const smhip_context* h, const NdtHost& n, int slot) {
  const NdtSlotMeta& m = n.meta[slot];
  return h->target_cache && m.valid && m.gen == h->tgt_gen[slot] && m.resolution == n.opts.resolution &&
         m.min_points == n.opts.min_points_per_voxel && m.eig_mult == n.opts.min_covar_eigvalue_mult;
}

// per-slot sizes / options into the device array the kernels index (one small copy per Align)
smhip_status ndt_push_devs(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  int off = 0;
  for (int k = first; k < first + K; ++k) {
    NdtDev& d = n.devs_host[k];
    d.nt = h->nt[k]; d.ns = h->ns[k];
    d.min_points = n.opts.min_points_per_voxel; d.eig_mult = n.opts.min_covar_eigvalue_mult;
    d.key_off = off; off += d.nt;
  }
  HIPCHK(h, hipMemcpyAsync(n.devs_dev + first, n.devs_host + first, sizeof(NdtDev) * K, hipMemcpyHostToDevice, h->stream));
  return SMHIP_OK;
}

// VoxelGridCovariance::filter(true) on the targets of slots [first, first + K) (ndt_omp.h:117-122 -> init()): every kernel once
// for the whole batch (grid.y = table), ONE radix sort of all the (table, voxel code, point) keys
smhip_status ndt_build_grids(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  smhip_status s = ndt_push_devs(h, first, K);
  if (s) return s;
  int nt_max = 0, nt_sum = 0;
  for (int k = first; k < first + K; ++k) { nt_max = std::max(nt_max, h->nt[k]); nt_sum += h->nt[k]; n.meta[k].valid = false; }
  // tgt_reduce reads nt from the pair input block
  for (int k = first; k < first + K; ++k) { h->in_pinned[k].nt = h->nt[k]; h->in_pinned[k].ns = h->ns[k]; }
  HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(h->dev.in) + first, h->in_pinned + first, sizeof(PairInput) * K, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemsetAsync(n.bits_all + (size_t)first * kNdtMaxWords, 0, sizeof(uint32_t) * (size_t)kNdtMaxWords * K, h->stream));
  IcpDev rd = h->dev; rd.npairs = K; rd.pair_base = first;
  const NdtDev* devs = n.devs_dev + first;
  const int gb = ceil_div(nt_max, 256);
  hipLaunchKernelGGL(tgt_reduce, dim3(kTgtReduceBlocks, K), dim3(256), 0, h->stream, rd);
  hipLaunchKernelGGL(ndt_voxel_setup, dim3(K), dim3(64), 0, h->stream, devs, n.opts.resolution);
  // (voxel code, point) pairs sorted with the rocPRIM radix sort of the workspace; the voxel's slot is the rank of its bit,
  // which is also its position among the sorted unique codes, so the sorted points ARE vpts
  hipLaunchKernelGGL(ndt_voxel_keys64, dim3(gb, K), dim3(256), 0, h->stream, devs, prep_keys(n.prep, 0), prep_values(n.prep, 0));
  int kbits = 0;
  while ((1 << kbits) < K) ++kbits;
  const hipError_t e = prep_sort_pairs(n.prep, h->stream, nt_sum, 33 + kbits);
  if (e != hipSuccess) { h->err = std::string("NDT voxel sort: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  hipLaunchKernelGGL(ndt_voxel_heads, dim3(gb, K), dim3(256), 0, h->stream, devs, prep_keys(n.prep, 1));
  hipLaunchKernelGGL(ndt_voxel_rank, dim3(K), dim3(1024), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_starts, dim3(gb, K), dim3(256), 0, h->stream, devs, prep_keys(n.prep, 1), prep_values(n.prep, 1));
  // one wave per occupied voxel; nocc <= nt
  hipLaunchKernelGGL(ndt_voxel_stats, dim3(std::min(ceil_div(nt_max, 4), std::max(64, 8192 / K)), K), dim3(256), 0, h->stream, devs);
  HIPCHK(h, hipMemcpyAsync(n.info_pinned + first, n.info_all + first, sizeof(NdtGridInfo) * K, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  for (int k = first; k < first + K; ++k) {
    if (n.info_pinned[k].status) { h->err = "NDT voxel box exceeds the bit grid (leaf size too small for the target extent)"; return SMHIP_ERR_CAPACITY; }
    NdtSlotMeta& m = n.meta[k];
    m.valid = true; m.gen = h->tgt_gen[k];
    m.resolution = n.opts.resolution; m.min_points = n.opts.min_points_per_voxel; m.eig_mult = n.opts.min_covar_eigvalue_mult;
  }
  return SMHIP_OK;
}

void ndt_fill_pose(const NdtHost& n, const double* p, const float* T, bool hess, NdtPose& P) {
  for (int i = 0; i < 12; ++i) P.T[i] = T[i];
  angle_derivatives(p, P);
  const double c1 = 10.0 * (1 - (double)n.opts.outlier_ratio);                     // :86-93
  const double c2 = (double)n.opts.outlier_ratio / std::pow((double)n.opts.resolution, 3);
  const double d3 = -std::log(c2);
  const double d1 = -std::log(c1 + c2) - d3;
  const double d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / d1);
  P.d1d = d1; P.d2d = d2; P.d1 = (float)d1; P.d2 = (float)d2;
  P.res2 = n.opts.resolution * n.opts.resolution;
  P.compute_hessian = hess ? 1 : 0;
}

// computeDerivatives (ndt_omp_impl.hpp:180-284) for `count` evaluations at once: evaluation e = poses_host[e] against the table of
// slot active_host[e]; results in out_pinned[e][0..43] (score, 6-gradient, 6x6 hessian, pair count)
smhip_status ndt_eval_round(smhip_context* h, int count, int ns_max) {
  NdtHost& n = ndt_of(h);
  // (workgroups per evaluation from the LARGEST source: a smaller cloud's surplus workgroups add zero rows, and the fold of the
  // rows is grouped by workgroup index, so a pair's sums are the bits its single call gives -- up to 524 288 source points)
  const int blocks = std::min(kNdtMaxDerivBlocks, std::max(1, ceil_div(ns_max, kNdtDerivThreads)));
  const bool one = (long long)blocks * kNdtDerivThreads >= ns_max;      // a thread per source point
  const dim3 g(blocks, count);
  if (count <= kNdtArgPoses) {
    // a small round (every round of a single Align): poses and tables ride in the launches' arguments, the sums come back through
    // page-locked memory -- no copy either way
    NdtPoseArgs A{};
    NdtActiveArgs S{};
    A.n = count;
    for (int e = 0; e < count; ++e) { A.active[e] = n.active_host[e]; S.slot[e] = n.active_host[e]; A.p[e] = n.poses_host[e]; }
    if (n.double_math) {
      if (one) hipLaunchKernelGGL((ndt_derivatives_args<double, true>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, A);
      else hipLaunchKernelGGL((ndt_derivatives_args<double, false>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, A);
    } else {
      if (one) hipLaunchKernelGGL((ndt_derivatives_args<float, true>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, A);
      else hipLaunchKernelGGL((ndt_derivatives_args<float, false>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, A);
    }
    hipLaunchKernelGGL(ndt_reduce_args, dim3(count), dim3(16 * 64), 0, h->stream, n.devs_dev, S, blocks, n.out_pinned);
  } else {
    HIPCHK(h, hipMemcpyAsync(n.poses_dev, n.poses_host, sizeof(NdtPose) * count, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(n.active_dev, n.active_host, sizeof(int32_t) * count, hipMemcpyHostToDevice, h->stream));
    if (n.double_math) {
      if (one) hipLaunchKernelGGL((ndt_derivatives<double, true>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.poses_dev, n.active_dev);
      else hipLaunchKernelGGL((ndt_derivatives<double, false>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.poses_dev, n.active_dev);
    } else {
      if (one) hipLaunchKernelGGL((ndt_derivatives<float, true>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.poses_dev, n.active_dev);
      else hipLaunchKernelGGL((ndt_derivatives<float, false>), g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.poses_dev, n.active_dev);
    }
    hipLaunchKernelGGL(ndt_reduce, dim3(count), dim3(16 * 64), 0, h->stream, n.devs_dev, n.active_dev, blocks, n.out_pinned);
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  return SMHIP_OK;
}

double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }     // ndt_omp.h auxilaryFunction_PsiMT
double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {   // :633-670
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  else if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  else if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {   // :674-753
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
  } else if (std::fabs(g_t) <= std::fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double nxt = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? std::min(a_t + 0.66 * (a_u - a_t), nxt) : std::max(a_t + 0.66 * (a_u - a_t), nxt);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}


// ---- one Align as a state machine: it asks for one computeDerivatives evaluation at a time -----------------------------
// The control flow of NormalDistributionsTransform::computeTransformation (:81-171) and computeStepLengthMT (:757-916), cut at
// the evaluations: INIT (:119) -> per Newton iteration TRIAL (:809-813) -> MT (the line-search loop's evaluations, :870-878)
// -> HESS (:912-913 computeHessian, only after a line search that looped) -> next iteration.
struct NdtJob {
  enum Phase { kInit, kTrial, kMt, kHess, kDone };
  Phase phase = kInit;
  int slot = 0;
  double p[6], x_t[6], dir[6];
  float Tf[16];                    // the pose matrix of the last evaluation asked for = final_transformation_ at the end
  double sc = 0, g[6], H[36];
  int it = 0, deriv_calls = 0;
  double last_pairs = 0;
  // computeStepLengthMT's locals
  double phi_0 = 0, d_phi_0 = 0, a_t = 0, a_l = 0, a_u = 0, f_l = 0, g_l = 0, f_u = 0, g_u = 0;
  double phi_t = 0, d_phi_t = 0, psi_t = 0, d_psi_t = 0, step_max = 0, step_min = 0;
  bool interval_converged = false, open_interval = true;
  int step_iterations = 0;
  // the evaluation wanted next
  double eval_p[6];
  bool eval_hess = true;
};

constexpr double kMtMu = 1.e-4, kMtNu = 0.9;

void ndt_job_request(NdtJob& j, const double* x, bool hess, NdtJob::Phase ph) {
  for (int i = 0; i < 6; ++i) j.eval_p[i] = x[i];
  j.eval_hess = hess;
  j.phase = ph;
}

void ndt_job_newton(NdtJob& j, const smhip_ndt_options& o);

void ndt_job_finish_iteration(NdtJob& j, const smhip_ndt_options& o) {
  const double dp_norm = j.a_t;                                                    // :142
  for (int i = 0; i < 6; ++i) j.p[i] += j.dir[i] * dp_norm;                        // :143, :152
  const bool converged = j.it > o.max_iterations || (j.it && std::fabs(dp_norm) < o.transformation_epsilon);   // :158-162
  j.it++;                                                                          // :164
  if (converged) { j.phase = NdtJob::kDone; return; }
  ndt_job_newton(j, o);
}

// the line-search loop's head (:867): another trial value, the closing Hessian, or the end of the iteration
void ndt_job_mt_continue(NdtJob& j, const smhip_ndt_options& o) {
  if (!j.interval_converged && j.step_iterations < 10 && !(j.psi_t <= 0 && j.d_phi_t <= -kMtNu * j.d_phi_0)) {
    j.a_t = j.open_interval ? trial_value_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.psi_t, j.d_psi_t)
                            : trial_value_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.phi_t, j.d_phi_t);
    j.a_t = std::max(std::min(j.a_t, j.step_max), j.step_min);
    for (int i = 0; i < 6; ++i) j.x_t[i] = j.p[i] + j.dir[i] * j.a_t;
    pose_to_matrix_f32(j.x_t, j.Tf);
    ndt_job_request(j, j.x_t, false, NdtJob::kMt);
    return;
  }
  if (j.step_iterations) { ndt_job_request(j, j.x_t, true, NdtJob::kHess); return; }       // :912-913
  ndt_job_finish_iteration(j, o);
}

// one Newton iteration up to its first evaluation (:121-141, :757-813)
void ndt_job_newton(NdtJob& j, const smhip_ndt_options& o) {
  for (;;) {
    double mg[6], dp[6];
    for (int i = 0; i < 6; ++i) mg[i] = -j.g[i];
    svd_solve6(j.H, mg, dp);                                                       // :127-129
    double dp_norm = 0;
    for (int i = 0; i < 6; ++i) dp_norm += dp[i] * dp[i];
    dp_norm = std::sqrt(dp_norm);
    if (dp_norm == 0 || dp_norm != dp_norm) { j.phase = NdtJob::kDone; return; }   // :134-139
    for (int i = 0; i < 6; ++i) j.dir[i] = dp[i] / dp_norm;                        // :141
    // ---- computeStepLengthMT(p, dir, dp_norm, step_size, trans_eps / 2, ...) :757-916
    const double step_init = dp_norm;
    j.step_max = o.step_size; j.step_min = o.transformation_epsilon / 2;
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
      j.interval_converged = (j.step_max - j.step_min) > 0;                        // :795 (sic: the loop never runs with the wrapper's settings)
      j.open_interval = true;
      j.step_iterations = 0;
      j.a_t = std::max(std::min(step_init, j.step_max), j.step_min);
      for (int i = 0; i < 6; ++i) j.x_t[i] = j.p[i] + j.dir[i] * j.a_t;
      pose_to_matrix_f32(j.x_t, j.Tf);                                             // :803-806
      ndt_job_request(j, j.x_t, true, NdtJob::kTrial);                             // :809-813
      return;
    }
    // d_phi_0 == 0: no step; the iteration ends where it began
    const bool converged = j.it > o.max_iterations || (j.it && std::fabs(j.a_t) < o.transformation_epsilon);
    j.it++;
    if (converged) { j.phase = NdtJob::kDone; return; }
  }
}

void ndt_job_start(NdtJob& j, int slot, const double* guess_cm) {
  j = NdtJob{};
  j.slot = slot;
  // guess.cast<float>() (ndt.cc:58); final_transformation_ = guess (:98)
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) j.Tf[4 * r + c] = (float)guess_cm[4 * c + r];
  float e[3];
  euler_xyz_f32(j.Tf, e);
  j.p[0] = j.Tf[3]; j.p[1] = j.Tf[7]; j.p[2] = j.Tf[11]; j.p[3] = e[0]; j.p[4] = e[1]; j.p[5] = e[2];   // :107-111
  ndt_job_request(j, j.p, true, NdtJob::kInit);                                    // :119
}

// the evaluation the job asked for has come back: out = score, gradient, hessian, pair count
void ndt_job_result(NdtJob& j, const double* out, const smhip_ndt_options& o) {
  j.deriv_calls++;
  j.last_pairs = out[43];
  switch (j.phase) {
    case NdtJob::kInit:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];
      ndt_job_newton(j, o);
      break;
    case NdtJob::kTrial:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];
      j.phi_t = -j.sc; j.d_phi_t = 0;
      for (int i = 0; i < 6; ++i) j.d_phi_t -= j.g[i] * j.dir[i];
      j.psi_t = psi_mt(j.a_t, j.phi_t, j.phi_0, j.d_phi_0, kMtMu); j.d_psi_t = dpsi_mt(j.d_phi_t, j.d_phi_0, kMtMu);
      ndt_job_mt_continue(j, o);
      break;
    case NdtJob::kMt:
      j.sc = out[0];
      for (int i = 0; i < 6; ++i) j.g[i] = out[1 + i];                             // (no Hessian in the loop, :872)
      j.phi_t = -j.sc; j.d_phi_t = 0;
      for (int i = 0; i < 6; ++i) j.d_phi_t -= j.g[i] * j.dir[i];
      j.psi_t = psi_mt(j.a_t, j.phi_t, j.phi_0, j.d_phi_0, kMtMu); j.d_psi_t = dpsi_mt(j.d_phi_t, j.d_phi_0, kMtMu);
      if (j.open_interval && (j.psi_t <= 0 && j.d_psi_t >= 0)) {
        j.open_interval = false;
        j.f_l = j.f_l + j.phi_0 - kMtMu * j.d_phi_0 * j.a_l; j.g_l = j.g_l + kMtMu * j.d_phi_0;
        j.f_u = j.f_u + j.phi_0 - kMtMu * j.d_phi_0 * j.a_u; j.g_u = j.g_u + kMtMu * j.d_phi_0;
      }
      j.interval_converged = j.open_interval ? update_interval_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.psi_t, j.d_psi_t)
                                             : update_interval_mt(j.a_l, j.f_l, j.g_l, j.a_u, j.f_u, j.g_u, j.a_t, j.phi_t, j.d_phi_t);
      j.step_iterations++;
      ndt_job_mt_continue(j, o);
      break;
    case NdtJob::kHess:
      for (int i = 0; i < 36; ++i) j.H[i] = out[7 + i];                            // the Hessian alone: score and gradient stay the loop's last
      ndt_job_finish_iteration(j, o);
      break;
    case NdtJob::kDone:
      break;
  }
}

// pcl::Registration::getFitnessScore(): mean squared 1-NN distance of each slot's source, moved by its T (column-major 4x4), to
// the slot's raw target (ndt.cc:60, ndt_gicp.cc:88,101), for slots [first, first + K) in one pass
smhip_status fitness_scores(smhip_context* h, int first, int K, const double* T, double* out) {
  NdtHost& n = ndt_of(h);
  int ns_max = 0, nt_max = 0;
  std::vector<int> had(K);
  for (int k = 0; k < K; ++k) { had[k] = h->has_normals[first + k]; h->has_normals[first + k] = 1; }
  smhip_status s = fill_inputs(h, K, T, &ns_max, &nt_max, first);
  for (int k = 0; k < K; ++k) h->has_normals[first + k] = had[k];
  if (s) return s;
  // distances only: no tie-order requirement (skip the per-cell sort) and no previous match to seed a
  // ball search -> plain exact ring search (r = 1 certifies almost every query against a dense submap)
  const int sort_was = h->dev.sort_cells, ball_was = h->dev.use_ball;
  const int ring_was = h->dev.max_ring;
  h->dev.sort_cells = 0; h->dev.use_ball = 0;
  h->dev.max_ring = std::max(ring_was, 32);      // wide rings are cheap with the row-occupancy bitmap; fewer queries reach the brute-force sweep
  if (K == 1) {
    s = enqueue_prepare_one(h, first, nt_max);
    if (s == SMHIP_OK) s = enqueue_find_closests_half(h, whole_batch(h, 1, first), ns_max, 0);
  } else {
    // the search structure over the raw targets: kept like the voxel tables while every slot's target is unchanged
    bool cached = true;
    for (int k = first; k < first + K; ++k) cached = cached && grid_cached(h, k);
    const Half f = whole_batch(h, K, first);
    if (cached) {
      HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(f.d.in) + first, h->in_pinned + first, sizeof(PairInput) * K, hipMemcpyHostToDevice, h->stream));
      s = ensure_packed(h, first, K);
      hipLaunchKernelGGL(reset_scratch_light, dim3(std::min(1024, 8 * K)), dim3(256), 0, h->stream, f.d, first, K);
      hipLaunchKernelGGL(pose_setup, dim3(ceil_div(K, 64)), dim3(64), 0, h->stream, f.d, K);
      h->cache_hits++;
    } else {
      s = enqueue_resets(h, K, first);
      if (s == SMHIP_OK) s = enqueue_grid_build(h, f, nt_max);
    }
    if (s == SMHIP_OK) s = enqueue_find_closests_half(h, f, ns_max, 0);
  }
  h->dev.sort_cells = sort_was; h->dev.use_ball = ball_was; h->dev.max_ring = ring_was;
  if (s) return s;
  for (int k0 = 0; k0 < K; k0 += kFitnessArgPairs) {
    FitnessArgs A{};
    const int kn = std::min(kFitnessArgPairs, K - k0);
    for (int k = 0; k < kn; ++k) A.ns[k] = h->ns[first + k0 + k];
    hipLaunchKernelGGL(fitness_partial, dim3(64, kn), dim3(256), 0, h->stream, h->dev.d2, (size_t)h->dev.ns_cap, first + k0, A, n.fit_pinned + (size_t)128 * k0);
  }
  HIPCHK(h, hipMemsetAsync(h->dev.hist + (size_t)first * kHistBins, 0, sizeof(uint32_t) * kHistBins * (size_t)K, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < K; ++k) {
    double ssum = 0, cnt = 0;
    for (int b = 0; b < 64; ++b) { ssum += n.fit_pinned[128 * k + 2 * b]; cnt += n.fit_pinned[128 * k + 2 * b + 1]; }
    out[k] = cnt > 0 ? ssum / cnt : 1.7976931348623157e308;
  }
  h->ev_used = 0;
  return SMHIP_OK;
}
smhip_status fitness_score(smhip_context* h, const double* T, double* out) { return fitness_scores(h, 0, 1, T, out); }

// K Aligns in lock-step on slots [first, first + K)
smhip_status ndt_align_slots(smhip_context* h, int first, int K, const double* guesses, double* results, double* scores, smhip_ndt_stats* stats) {
  for (int k = first; k < first + K; ++k)
    if (h->ns[k] <= 0 || h->nt[k] <= 0) { h->err = "Ndt::Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }   // ndt.cc:40-42
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = ndt_ensure(h, first + K);
  if (s) return s;
  NdtHost& n = ndt_of(h);
  // setInputTarget -> init() on every Align (ndt.cc:54).  The voxel table is a pure function of the target and the
  // options, so it is kept while the slot's target is unchanged (smhip_set_target_cache(h, 0) = rebuild every time); a batch
  // keeps its tables when every one of them is current, else rebuilds them all in one pass
  bool current = true;
  for (int k = first; k < first + K; ++k) current = current && ndt_table_current(h, n, k);
  if (current) {
    s = ndt_push_devs(h, first, K);
    if (s) return s;
    h->cache_hits++;
  } else {
    s = ndt_build_grids(h, first, K);
    if (s) return s;
  }
  const smhip_ndt_options& o = n.opts;
  int ns_max = 0;
  for (int k = first; k < first + K; ++k) ns_max = std::max(ns_max, h->ns[k]);
  std::vector<NdtJob> jobs(K);
  for (int k = 0; k < K; ++k) ndt_job_start(jobs[k], first + k, guesses + 16 * k);
  std::vector<int> who(K);
  for (;;) {
    int count = 0;
    for (int k = 0; k < K; ++k) {
      NdtJob& j = jobs[k];
      if (j.phase == NdtJob::kDone) continue;
      ndt_fill_pose(n, j.eval_p, j.Tf, j.eval_hess, n.poses_host[count]);
      n.active_host[count] = j.slot;
      who[count++] = k;
    }
    if (!count) break;
    s = ndt_eval_round(h, count, ns_max);
    if (s) return s;
    for (int e = 0; e < count; ++e) ndt_job_result(jobs[who[e]], n.out_pinned + (size_t)e * kNdtDerivCols, o);
  }
  // getFinalTransformation().cast<double>(), column-major out (ndt.cc:61)
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) results[16 * k + 4 * c + r] = (double)jobs[k].Tf[4 * r + c];
  std::vector<double> fit(K);
  s = fitness_scores(h, first, K, results, fit.data());
  if (s) return s;
  for (int k = 0; k < K; ++k) {
    if (scores) scores[k] = fit[k];
    if (stats) {
      stats[k].iterations = jobs[k].it;
      stats[k].derivative_calls = jobs[k].deriv_calls;
      stats[k].voxels = n.info_pinned[first + k].nocc;
      stats[k].status = 0;
      stats[k].trans_probability = jobs[k].sc / (double)h->ns[first + k];      // :170
      stats[k].pairs_last = jobs[k].last_pairs;
    }
  }
  n.deriv_calls = jobs[0].deriv_calls; n.last_pairs = jobs[0].last_pairs;
  return SMHIP_OK;
}

}  // namespace

extern "C" {

void smhip_ndt_default_options(smhip_ndt_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->resolution = 1.0f;               // ndt.cc:31
  o->step_size = 0.1f;                // ndt_omp_impl.hpp:50
  o->outlier_ratio = 0.55f;           // :51
  o->transformation_epsilon = 0.1f;   // :71
  o->max_iterations = 35;             // :72
  o->min_points_per_voxel = 6;        // voxel_grid_covariance_omp.h:204
  o->min_covar_eigvalue_mult = 0.01f; // :205
}

smhip_status smhip_ndt_set_options(smhip_handle h, const smhip_ndt_options* o) {
  if (!h || !o) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!(o->resolution > 0) || !(o->step_size > 0) || o->max_iterations < 0 || o->min_points_per_voxel < 3) {
    h->err = "bad NDT options";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  NdtHost& n = ndt_of(h);
  n.opts = *o;
  for (auto& m : n.meta) m.valid = false;
  return SMHIP_OK;
}

smhip_status smhip_ndt_align(smhip_handle h, const double guess[16], double result[16], double* score, smhip_ndt_stats* stats) {
  if (!h || !guess || !result) return SMHIP_ERR_INVALID_ARGUMENT;
  return ndt_align_slots(h, 0, 1, guess, result, score, stats);
}

smhip_status smhip_ndt_align_batch(smhip_handle h, int first_slot, int npairs, const double* guesses, double* results, double* scores, smhip_ndt_stats* stats) {
  if (!h || !guesses || !results || npairs < 1 || first_slot < 0 || first_slot + npairs > h->dev.slots) {
    if (h) h->err = "bad slot range / guesses";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  return ndt_align_slots(h, first_slot, npairs, guesses, results, scores, stats);
}

// Test hooks: build the voxel grid of slot 0's target / evaluate computeDerivatives at a pose.
smhip_status smhip_ndt_build_voxels(smhip_handle h, int* n_voxels) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (h->nt[0] <= 0) { h->err = "target not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = ndt_ensure(h, 1);
  if (s) return s;
  s = ndt_build_grids(h, 0, 1);
  if (s) return s;
  if (n_voxels) *n_voxels = ndt_of(h).info_pinned[0].nocc;
  return SMHIP_OK;
}

smhip_status smhip_ndt_get_voxels(smhip_handle h, int capacity, int32_t* keys, int32_t* counts, double* means, float* icovs, float* centroids) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < 1 || !n.meta[0].valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  const int nocc = n.info_pinned[0].nocc;
  if (capacity < nocc) { h->err = "capacity too small"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(ndt_voxel_keys, dim3(ceil_div(nocc, 256)), dim3(256), 0, h->stream, n.devs_dev, n.vkey);
  std::vector<NdtVoxel> vox(nocc);
  std::vector<int32_t> k(nocc);
  HIPCHK(h, hipMemcpyAsync(vox.data(), n.vox_all, sizeof(NdtVoxel) * nocc, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(k.data(), n.vkey, sizeof(int32_t) * nocc, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int v = 0; v < nocc; ++v) {
    if (keys) keys[v] = k[v];
    if (counts) counts[v] = vox[v].n;
    if (means) for (int c = 0; c < 3; ++c) means[3 * v + c] = vox[v].mean[c];
    if (icovs) for (int c = 0; c < 6; ++c) icovs[6 * v + c] = vox[v].icov[c];
    if (centroids) for (int c = 0; c < 3; ++c) centroids[3 * v + c] = vox[v].centroid[c];
  }
  return SMHIP_OK;
}

smhip_status smhip_ndt_compute_derivatives(smhip_handle h, const double pose6[6], int compute_hessian, double* score, double grad[6], double hess[36]) {
  if (!h || !pose6 || !score || !grad || !hess) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < 1 || !n.meta[0].valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  if (h->ns[0] <= 0) { h->err = "source not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = ndt_push_devs(h, 0, 1);
  if (s) return s;
  float Tf[16];
  pose_to_matrix_f32(pose6, Tf);
  ndt_fill_pose(n, pose6, Tf, compute_hessian != 0, n.poses_host[0]);
  n.active_host[0] = 0;
  s = ndt_eval_round(h, 1, h->ns[0]);
  if (s) return s;
  *score = n.out_pinned[0];
  for (int i = 0; i < 6; ++i) grad[i] = n.out_pinned[1 + i];
  for (int i = 0; i < 36; ++i) hess[i] = compute_hessian ? n.out_pinned[7 + i] : 0.0;
  n.last_pairs = n.out_pinned[43];
  n.deriv_calls++;
  return SMHIP_OK;
}

}  // extern "C"
