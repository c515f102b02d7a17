// smhip_ndt_api.hip -- host side of registrators::Ndt on the C ABI (included by smhip_api.hip).
//
// Ndt::Align (/root/reference/registrators/ndt.cc:38-64) = convert clouds, setInputTarget (voxel grid
// build, every call), pclomp NDT align, getFitnessScore.  The 6-vector Newton / More-Thuente driver
// (pclomp/ndt_omp_impl.hpp:81-171, 757-916) runs here on the host exactly as in the reference; each
// computeDerivatives call is one ndt_derivatives + ndt_reduce launch and a 44-double read-back.
//
// Provenance note: computeStepLengthMT / trialValueSelectionMT / updateIntervalMT below are a PORT, not a redesign -- ~80
// lines of scalar host control flow that follow pclomp/ndt_omp_impl.hpp:633-916 branch for branch (same variable roles:
// a_l, f_l, g_l, a_u, phi_0, d_psi_t, open_interval ...), including the reference's quirk that `interval_converged` is
// computed from the UN-updated interval.  The iteration and derivative-call counts of the parity tests depend on that control
// flow being reproduced exactly; nothing in it is data-parallel.  Everything it calls (the derivative evaluations, the voxel
// table, the fitness search) is this repository's own GPU code.
#include "ndt_kernels.hip"

namespace {

struct NdtHost {
  NdtDev dev{};
  bool allocated = false;
  bool grid_valid = false;
  unsigned long long grid_gen = 0;    // tgt_gen[0] the voxel grid was built from
  float grid_resolution = 0.f; int grid_min_points = 0; float grid_eig_mult = 0.f;   // ... and the options it depends on
  smhip_ndt_options opts{};
  double* out_pinned = nullptr;       // kNdtDerivCols doubles
  NdtGridInfo* info_pinned = nullptr;
  double* fit_dev = nullptr;          // 2 * 64 doubles
  double* fit_pinned = nullptr;
  int32_t* vkey = nullptr;            // [nt] linear voxel index of every occupied voxel (tests)
  int deriv_calls = 0;
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

smhip_status ndt_ensure(smhip_context* h) {
  NdtHost& n = ndt_of(h);
  if (n.allocated) return SMHIP_OK;
  const size_t NT = h->dev.nt_cap;
  NdtDev& d = n.dev;
  smhip_status s = SMHIP_OK;
  auto A = [&](smhip_status r) { if (s == SMHIP_OK) s = r; };
  A(dev_alloc(h, &d.info, 1));
  A(dev_alloc(h, &d.bits, (size_t)kNdtMaxWords));
  A(dev_alloc(h, &d.words, (size_t)kNdtMaxWords));
  A(dev_alloc(h, &d.vstart, NT + 1));
  A(dev_alloc(h, &d.vpts, NT));
  A(dev_alloc(h, &d.vox, NT));
  A(dev_alloc(h, &d.icovd, NT * 6));
  A(dev_alloc(h, &d.partials, (size_t)kNdtMaxDerivBlocks * kNdtDerivCols));
  A(dev_alloc(h, &d.out, (size_t)kNdtDerivCols));
  A(dev_alloc(h, &n.fit_dev, 128));
  A(dev_alloc(h, &n.vkey, NT));
  if (s) return s;
  if (hipHostMalloc(reinterpret_cast<void**>(&n.out_pinned), sizeof(double) * kNdtDerivCols) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&n.info_pinned), sizeof(NdtGridInfo)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&n.fit_pinned), sizeof(double) * 128) != hipSuccess) {
    h->err = "hipHostMalloc failed (NDT)";
    return SMHIP_ERR_HIP;
  }
  n.allocated = true;
  return SMHIP_OK;
}

__global__ void ndt_voxel_keys(NdtDev d, int32_t* keys) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.info->nocc) return;
  const float4 p = d.vpts[d.vstart[v]];
  int i0, i1, i2;
  ndt_voxel_of(d.info, p.x, p.y, p.z, i0, i1, i2);
  keys[v] = i0 + i1 * d.info->div_b[0] + i2 * d.info->div_b[0] * d.info->div_b[1];   // :223
}

// VoxelGridCovariance::filter(true) on the slot-0 target (ndt_omp.h:117-122 -> init())
smhip_status ndt_build_grid(smhip_context* h) {
  NdtHost& n = ndt_of(h);
  NdtDev& d = n.dev;
  d.nt = h->nt[0]; d.ns = h->ns[0];
  d.tgt = h->dev.tgt_p; d.src = h->dev.src; d.tpart = h->dev.tpart;
  d.min_points = n.opts.min_points_per_voxel;
  d.eig_mult = n.opts.min_covar_eigvalue_mult;
  // tgt_reduce reads nt from the pair input block
  h->in_pinned[0].nt = h->nt[0]; h->in_pinned[0].ns = h->ns[0];
  HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(h->dev.in), h->in_pinned, sizeof(PairInput), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemsetAsync(d.bits, 0, sizeof(uint32_t) * (size_t)kNdtMaxWords, h->stream));
  const int gb = ceil_div(d.nt, 256);
  hipLaunchKernelGGL(tgt_reduce, dim3(kTgtReduceBlocks, 1), dim3(256), 0, h->stream, h->dev);
  hipLaunchKernelGGL(ndt_voxel_setup, dim3(1), dim3(64), 0, h->stream, d, n.opts.resolution);
  // (voxel code, point) pairs sorted with the rocPRIM radix sort of the preparation workspace; the voxel's slot is the
  // rank of its bit, which is also its position among the sorted unique codes, so the sorted points ARE vpts
  {
    smhip_status ps = prep_ensure(h);
    if (ps) return ps;
    hipLaunchKernelGGL(ndt_voxel_keys64, dim3(gb), dim3(256), 0, h->stream, d, prep_keys(h->prep, 0), prep_values(h->prep, 0));
    const hipError_t e = prep_sort_pairs(h->prep, h->stream, d.nt, 33);
    if (e != hipSuccess) { h->err = std::string("NDT voxel sort: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
    hipLaunchKernelGGL(ndt_voxel_heads, dim3(gb), dim3(256), 0, h->stream, d, prep_keys(h->prep, 1));
    hipLaunchKernelGGL(ndt_voxel_rank, dim3(1), dim3(1024), 0, h->stream, d);
    hipLaunchKernelGGL(ndt_voxel_starts, dim3(gb), dim3(256), 0, h->stream, d, prep_keys(h->prep, 1), prep_values(h->prep, 1));
  }
  // one wave per occupied voxel; nocc <= nt
  hipLaunchKernelGGL(ndt_voxel_stats, dim3(ceil_div(d.nt, 4)), dim3(256), 0, h->stream, d);
  HIPCHK(h, hipMemcpyAsync(n.info_pinned, d.info, sizeof(NdtGridInfo), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  if (n.info_pinned->status) { h->err = "NDT voxel box exceeds the bit grid (leaf size too small for the target extent)"; return SMHIP_ERR_CAPACITY; }
  n.grid_valid = true;
  n.grid_gen = h->tgt_gen[0];
  n.grid_resolution = n.opts.resolution; n.grid_min_points = n.opts.min_points_per_voxel; n.grid_eig_mult = n.opts.min_covar_eigvalue_mult;
  return SMHIP_OK;
}

// computeDerivatives (ndt_omp_impl.hpp:180-284): score, 6-gradient, 6x6 hessian
smhip_status ndt_derivs(smhip_context* h, const double* p, const float* T, bool hess, double* score, double* g, double* H) {
  NdtHost& n = ndt_of(h);
  NdtPose P{};
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
  const int blocks = std::min(kNdtMaxDerivBlocks, std::max(1, ceil_div(n.dev.ns, kNdtDerivThreads)));
  const bool one = (long long)blocks * kNdtDerivThreads >= n.dev.ns;     // a thread per source point
  if (n.double_math) {
    if (one) hipLaunchKernelGGL((ndt_derivatives<double, true>), dim3(blocks), dim3(kNdtDerivThreads), 0, h->stream, n.dev, P);
    else hipLaunchKernelGGL((ndt_derivatives<double, false>), dim3(blocks), dim3(kNdtDerivThreads), 0, h->stream, n.dev, P);
  } else {
    if (one) hipLaunchKernelGGL((ndt_derivatives<float, true>), dim3(blocks), dim3(kNdtDerivThreads), 0, h->stream, n.dev, P);
    else hipLaunchKernelGGL((ndt_derivatives<float, false>), dim3(blocks), dim3(kNdtDerivThreads), 0, h->stream, n.dev, P);
  }
  hipLaunchKernelGGL(ndt_reduce, dim3(1), dim3(16 * 64), 0, h->stream, n.dev, blocks);
  HIPCHK(h, hipMemcpyAsync(n.out_pinned, n.dev.out, sizeof(double) * kNdtDerivCols, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  *score = n.out_pinned[0];
  for (int i = 0; i < 6; ++i) g[i] = n.out_pinned[1 + i];
  if (hess) for (int i = 0; i < 36; ++i) H[i] = n.out_pinned[7 + i];
  n.last_pairs = n.out_pinned[43];
  n.deriv_calls++;
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

// pcl::Registration::getFitnessScore(): mean squared 1-NN distance of slot 0's source, moved by `T`
// (column-major 4x4), to slot 0's raw target (ndt.cc:60, ndt_gicp.cc:88,101).
smhip_status fitness_score(smhip_context* h, const double* T, double* out) {
  NdtHost& n = ndt_of(h);
  int ns_max = 0, nt_max = 0;
  const int had = h->has_normals[0];
  h->has_normals[0] = 1;
  smhip_status s = fill_inputs(h, 1, T, &ns_max, &nt_max);
  h->has_normals[0] = had;
  if (s) return s;
  // distances only: no tie-order requirement (skip the per-cell sort) and no previous match to seed a
  // ball search -> plain exact ring search (r = 1 certifies almost every query against a dense submap)
  const int sort_was = h->dev.sort_cells, ball_was = h->dev.use_ball;
  const int ring_was = h->dev.max_ring;
  h->dev.sort_cells = 0; h->dev.use_ball = 0;
  h->dev.max_ring = std::max(ring_was, 32);      // wide rings are cheap with the row-occupancy bitmap; fewer queries reach the brute-force sweep
  s = enqueue_prepare(h, 1, nt_max);
  if (s == SMHIP_OK) s = enqueue_find_closests(h, 1, ns_max);
  h->dev.sort_cells = sort_was; h->dev.use_ball = ball_was; h->dev.max_ring = ring_was;
  if (s) return s;
  hipLaunchKernelGGL(fitness_partial, dim3(64), dim3(256), 0, h->stream, h->dev.d2, h->ns[0], n.fit_dev);
  HIPCHK(h, hipMemsetAsync(h->dev.hist, 0, sizeof(uint32_t) * kHistBins, h->stream));
  HIPCHK(h, hipMemcpyAsync(n.fit_pinned, n.fit_dev, sizeof(double) * 128, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double ssum = 0, cnt = 0;
  for (int k = 0; k < 64; ++k) { ssum += n.fit_pinned[2 * k]; cnt += n.fit_pinned[2 * k + 1]; }
  *out = cnt > 0 ? ssum / cnt : 1.7976931348623157e308;
  h->ev_used = 0;
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
  ndt_of(h).opts = *o;
  ndt_of(h).grid_valid = false;
  return SMHIP_OK;
}

smhip_status smhip_ndt_align(smhip_handle h, const double guess[16], double result[16], double* score, smhip_ndt_stats* stats) {
  if (!h || !guess || !result) return SMHIP_ERR_INVALID_ARGUMENT;
  if (h->ns[0] <= 0 || h->nt[0] <= 0) { h->err = "Ndt::Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }   // ndt.cc:40-42
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = ndt_ensure(h);
  if (s) return s;
  NdtHost& n = ndt_of(h);
  n.deriv_calls = 0;
  // setInputTarget -> init() on every Align (ndt.cc:54).  The voxel table is a pure function of the target and the
  // options, so it is kept while the slot's target is unchanged (smhip_set_target_cache(h, 0) = rebuild every time)
  if (h->target_cache && n.grid_valid && n.grid_gen == h->tgt_gen[0] && n.grid_resolution == n.opts.resolution &&
      n.grid_min_points == n.opts.min_points_per_voxel && n.grid_eig_mult == n.opts.min_covar_eigvalue_mult) {
    n.dev.ns = h->ns[0]; n.dev.src = h->dev.src;
    h->cache_hits++;
  } else {
    s = ndt_build_grid(h);
    if (s) return s;
  }
  const smhip_ndt_options& o = n.opts;
  // guess.cast<float>() (ndt.cc:58); final_transformation_ = guess (:98)
  float Tf[16];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Tf[4 * r + c] = (float)guess[4 * c + r];
  double p[6];
  { float e[3]; euler_xyz_f32(Tf, e); p[0] = Tf[3]; p[1] = Tf[7]; p[2] = Tf[11]; p[3] = e[0]; p[4] = e[1]; p[5] = e[2]; }   // :107-111
  double sc = 0, g[6], H[36];
  s = ndt_derivs(h, p, Tf, true, &sc, g, H);               // :119
  if (s) return s;
  int it = 0;
  bool converged = false;
  while (!converged) {                                     // :121
    double mg[6], dp[6];
    for (int i = 0; i < 6; ++i) mg[i] = -g[i];
    svd_solve6(H, mg, dp);                                 // :127-129
    double dp_norm = 0;
    for (int i = 0; i < 6; ++i) dp_norm += dp[i] * dp[i];
    dp_norm = std::sqrt(dp_norm);
    if (dp_norm == 0 || dp_norm != dp_norm) break;         // :134-139
    double dir[6];
    for (int i = 0; i < 6; ++i) dir[i] = dp[i] / dp_norm;  // :141
    // ---- computeStepLengthMT(p, dir, dp_norm, step_size, trans_eps / 2, ...) :757-916
    const double step_init = dp_norm, step_max = o.step_size, step_min = o.transformation_epsilon / 2;
    const double phi_0 = -sc;
    double d_phi_0 = 0;
    for (int i = 0; i < 6; ++i) d_phi_0 -= g[i] * dir[i];
    double a_t = 0;
    bool skip = false;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) skip = true;
      else { d_phi_0 *= -1; for (int i = 0; i < 6; ++i) dir[i] = -dir[i]; }
    }
    if (!skip) {
      const double mu = 1.e-4, nu = 0.9;
      double a_l = 0, a_u = 0;
      double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
      double f_u = f_l, g_u = g_l;
      bool interval_converged = (step_max - step_min) > 0, open_interval = true;   // :795 (sic: the loop below never runs with the wrapper's settings)
      a_t = std::max(std::min(step_init, step_max), step_min);
      double x_t[6];
      for (int i = 0; i < 6; ++i) x_t[i] = p[i] + dir[i] * a_t;
      pose_to_matrix_f32(x_t, Tf);                         // :803-806
      s = ndt_derivs(h, x_t, Tf, true, &sc, g, H);         // :809-813
      if (s) return s;
      double phi_t = -sc, d_phi_t = 0;
      for (int i = 0; i < 6; ++i) d_phi_t -= g[i] * dir[i];
      double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
      int step_iterations = 0;
      while (!interval_converged && step_iterations < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        a_t = open_interval ? trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                            : trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = std::max(std::min(a_t, step_max), step_min);
        for (int i = 0; i < 6; ++i) x_t[i] = p[i] + dir[i] * a_t;
        pose_to_matrix_f32(x_t, Tf);
        double Hd[36];
        s = ndt_derivs(h, x_t, Tf, false, &sc, g, Hd);
        if (s) return s;
        phi_t = -sc; d_phi_t = 0;
        for (int i = 0; i < 6; ++i) d_phi_t -= g[i] * dir[i];
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu); d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          open_interval = false;
          f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
          f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        interval_converged = open_interval ? update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                           : update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
      }
      if (step_iterations) {                               // :912-913 computeHessian
        double sd, gd[6];
        s = ndt_derivs(h, x_t, Tf, true, &sd, gd, H);
        if (s) return s;
      }
    }
    dp_norm = a_t;                                         // :142
    for (int i = 0; i < 6; ++i) p[i] += dir[i] * dp_norm;  // :143, :152
    if (it > o.max_iterations || (it && std::fabs(dp_norm) < o.transformation_epsilon)) converged = true;   // :158-162
    it++;                                                  // :164
  }
  // getFinalTransformation().cast<double>(), column-major out (ndt.cc:61)
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) result[4 * c + r] = (double)Tf[4 * r + c];
  double fit = 0;
  s = fitness_score(h, result, &fit);
  if (s) return s;
  if (score) *score = fit;
  if (stats) {
    stats->iterations = it;
    stats->derivative_calls = n.deriv_calls;
    stats->voxels = n.info_pinned->nocc;
    stats->status = 0;
    stats->trans_probability = sc / (double)h->ns[0];      // :170
    stats->pairs_last = n.last_pairs;
  }
  return SMHIP_OK;
}

// Test hooks: build the voxel grid of slot 0's target / evaluate computeDerivatives at a pose.
smhip_status smhip_ndt_build_voxels(smhip_handle h, int* n_voxels) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (h->nt[0] <= 0) { h->err = "target not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = ndt_ensure(h);
  if (s) return s;
  s = ndt_build_grid(h);
  if (s) return s;
  if (n_voxels) *n_voxels = ndt_of(h).info_pinned->nocc;
  return SMHIP_OK;
}

smhip_status smhip_ndt_get_voxels(smhip_handle h, int capacity, int32_t* keys, int32_t* counts, double* means, float* icovs, float* centroids) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (!n.grid_valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  const int nocc = n.info_pinned->nocc;
  if (capacity < nocc) { h->err = "capacity too small"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(ndt_voxel_keys, dim3(ceil_div(nocc, 256)), dim3(256), 0, h->stream, n.dev, n.vkey);
  std::vector<NdtVoxel> vox(nocc);
  std::vector<int32_t> k(nocc);
  HIPCHK(h, hipMemcpyAsync(vox.data(), n.dev.vox, sizeof(NdtVoxel) * nocc, hipMemcpyDeviceToHost, h->stream));
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
  if (!n.grid_valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  if (h->ns[0] <= 0) { h->err = "source not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  n.dev.ns = h->ns[0];
  float Tf[16];
  pose_to_matrix_f32(pose6, Tf);
  for (int i = 0; i < 36; ++i) hess[i] = 0;
  return ndt_derivs(h, pose6, Tf, compute_hessian != 0, score, grad, hess);
}

}  // extern "C"
