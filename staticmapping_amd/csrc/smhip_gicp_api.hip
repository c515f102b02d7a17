// smhip_gicp_api.hip -- host side of registrators::NdtWithGicp on the C ABI (included by smhip_api.hip).
//
// NdtWithGicp::Align (/root/reference/registrators/ndt_gicp.cc:55-112) = pcl::ApproximateVoxelGrid on both clouds
// -> stock pcl NDT -> (fitness <= 1) stock pcl GICP -> score exp(-fitness).  All three live in PCL (1.8.1 pinned,
// see oracle/ndt_gicp.py); GICP statements are cited from the in-tree fork registrators/pclomp/gicp_omp_impl.hpp.
// The matcher keeps its raw clouds in private device buffers (the reference converts its stored clouds in every
// Align, ndt_gicp.cc:59-76).  A handle of S pair slots runs up to S / 2 JOBS side by side: job j works in pair slot j
// (down-sampled clouds, their search grid) and uses slot S / 2 + j as scratch for the source's own neighbour search, so the
// handle needs pair_slots >= 2.  The single Align is job 0; smhip_ndt_gicp_align_batch advances its jobs in lock-step: every
// job keeps the reference's own sequence of functor evaluations (its BFGS runs as a coroutine on the host, unchanged), and one
// round of the batch serves each live job's next request -- a correspondence step or a functor evaluation -- with one launch
// per kind over all of them and one hand-back through page-locked memory.
#include <chrono>
#include <thread>
#include <memory>
#include <cstdlib>
#include <cstdio>
#include <ucontext.h>
#include "gicp_kernels.hip"

namespace {

struct GicpJobHost {
  int n_raw_src = 0, n_raw_tgt = 0;
  // what is derived from the target alone is kept while the target is (smhip_set_target_cache, default on): the front end
  // aligns scan after scan against one submap, and ndt_gicp.cc filters / rebuilds / re-estimates all of it in every Align
  unsigned long long raw_tgt_gen = 0;       // bumped by every smhip_ndt_gicp_set_target_f32
  unsigned long long staged_raw_gen = 0;    // the raw target the job slot's staged target was made from ...
  unsigned long long staged_slot_gen = 0;   // ... and the slot's tgt_gen right after staging it
  int staged_filter = -1; float staged_res = 0.f; int staged_nt = 0;
  // target covariances are estimated when a point is first matched (gicp_need / gicp_knn_cov_listed); an entry is valid when its
  // stamp in cov_epoch equals `epoch`, which moves on whenever the target or the parameters change
  unsigned long long cov_gen = 0;           // the slot's tgt_gen the current epoch belongs to, with these parameters
  int cov_k = 0; double cov_eps = 0;
  uint32_t epoch = 0;
  bool cov_full = false;                    // every entry was estimated up front (a single Align does that: see gicp_align_jobs)
};

struct GicpHost {
  GicpDev dev{};
  bool allocated = false;
  smhip_ndt_gicp_options opts{};
  int jobs = 0;                       // pair_slots / 2
  std::vector<GicpJobHost> job;
  float4* raw_src = nullptr;          // [jobs][ns_cap] as handed over
  float4* raw_tgt = nullptr;          // [jobs][nt_cap]
  float4* ds_tmp = nullptr;           // [max(ns_cap, nt_cap)] filter output before the Morton ordering
  float4* ds_src = nullptr;           // [jobs][ns_cap] the same for every job's source at once (batched staging; allocated on first use)
  PrepWorkspace* prep_avg = nullptr;  // workspace of the batched voxel filter: every job's raw source + target at once (allocated on first use)
  double* out_pinned = nullptr;       // [jobs][kGicpCols] doubles, then the finished round's number
  uint32_t* count_pinned = nullptr;   // [jobs]
  int evals = 0;                      // functor evaluations of the last single-job run (parity hooks)
  unsigned long long seq = 0;         // evaluation rounds launched so far: the number gicp_fdf stores after the last job's sums
};

GicpHost& gicp_of(smhip_context* h);
inline int gicp_scratch_slot(const smhip_context* h, int job) { return h->dev.slots / 2 + job; }

smhip_status gicp_ensure(smhip_context* h) {
  GicpHost& g = gicp_of(h);
  if (g.allocated) return SMHIP_OK;
  if (h->dev.slots < 2) { h->err = "NdtWithGicp needs a handle created with pair_slots >= 2"; return SMHIP_ERR_CAPACITY; }
  const int J = h->dev.slots / 2;
  smhip_status s = ndt_ensure(h, J);
  if (s) return s;
  s = prep_ensure(h);
  if (s) return s;
  const size_t NS = h->dev.ns_cap, NT = h->dev.nt_cap;
  auto A = [&](smhip_status r) { if (s == SMHIP_OK) s = r; };
  A(dev_alloc(h, &g.dev.cov_s, (size_t)J * NS * 6));
  A(dev_alloc(h, &g.dev.cov_t, (size_t)J * NT * 6));
  A(dev_alloc(h, &g.dev.maha, (size_t)J * NS * 6));
  A(dev_alloc(h, &g.dev.qraw, (size_t)J * NS));
  A(dev_alloc(h, &g.dev.partials, (size_t)J * kGicpMaxBlocks * kGicpCols));
  A(dev_alloc(h, &g.dev.out, (size_t)J * kGicpCols));
  A(dev_alloc(h, &g.dev.count, (size_t)J));
  A(dev_alloc(h, &g.dev.ticket, (size_t)J));
  A(dev_alloc(h, &g.dev.round_done, 4));
  A(dev_alloc(h, &g.dev.cov_epoch, (size_t)J * NT));
  A(dev_alloc(h, &g.dev.need_list, (size_t)J * NS));
  A(dev_alloc(h, &g.dev.need_count, (size_t)J));
  A(dev_alloc(h, &g.raw_src, (size_t)J * NS));
  A(dev_alloc(h, &g.raw_tgt, (size_t)J * NT));
  A(dev_alloc(h, &g.ds_tmp, std::max(NS, NT)));
  if (s) return s;
  if (hipHostMalloc(reinterpret_cast<void**>(&g.out_pinned), sizeof(double) * ((size_t)J * kGicpCols + 1)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&g.count_pinned), sizeof(uint32_t) * (size_t)J) != hipSuccess) {
    h->err = "hipHostMalloc failed (GICP)";
    return SMHIP_ERR_HIP;
  }
  std::memset(g.out_pinned, 0, sizeof(double) * ((size_t)J * kGicpCols + 1));
  g.dev.out_host = g.out_pinned;                       // (page-locked host memory is device-addressable at the same pointer)
  g.dev.jobs = J;
  if (hipMemsetAsync(g.dev.cov_epoch, 0, sizeof(uint32_t) * (size_t)J * NT, h->stream) != hipSuccess ||
      hipMemsetAsync(g.dev.ticket, 0, sizeof(uint32_t) * (size_t)J, h->stream) != hipSuccess ||
      hipMemsetAsync(g.dev.round_done, 0, sizeof(uint32_t) * 4, h->stream) != hipSuccess) { h->err = "hipMemsetAsync failed (GICP)"; return SMHIP_ERR_HIP; }
  g.jobs = J;
  g.job.assign((size_t)J, GicpJobHost{});
  g.allocated = true;
  return SMHIP_OK;
}

// ---- 6-vector helpers -------------------------------------------------------------------------
// applyState (:516-527) on a float 4x4 (row-major): t.topLeft = Rz(x5) Ry(x4) Rx(x3) * t.topLeft ; t.col(3) += x0..2
void apply_state_f32(const float* Tin, const double* x, float* Tout) {
  const float a = (float)x[3], b = (float)x[4], c = (float)x[5];
  const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  const float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  const float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float M[9], R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += Rz[3 * i + k] * Ry[3 * k + j]; M[3 * i + j] = s; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += M[3 * i + k] * Rx[3 * k + j]; R[3 * i + j] = s; }
  float out[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += R[3 * i + k] * Tin[4 * k + j]; out[4 * i + j] = s; }
    out[4 * i + 3] = Tin[4 * i + 3] + (float)x[i];
  }
  out[12] = Tin[12]; out[13] = Tin[13]; out[14] = Tin[14]; out[15] = Tin[15];
  for (int i = 0; i < 16; ++i) Tout[i] = out[i];
}

// computeRDerivative (:133-186): g[3..5] = <dR/dphi, R>, <dR/dtheta, R>, <dR/dpsi, R>, <A, B> = sum A(j,i) B(i,j)
void r_derivative(const double* x, const double* R /*row-major 3x3*/, double* g) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = std::cos(phi), sphi = std::sin(phi), cth = std::cos(theta), sth = std::sin(theta);
  const double cpsi = std::cos(psi), spsi = std::sin(psi);
  const double dphi[9] = {0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth,
                          0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth,
                          0, cphi * cth, -cth * sphi};
  const double dth[9] = {-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth,
                         -spsi * sth, cth * sphi * spsi, cphi * cth * spsi,
                         -cth, -sphi * sth, -cphi * sth};
  const double dpsi[9] = {-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth,
                          cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth,
                          0, 0, 0};
  auto ip = [&](const double* A) { double r = 0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r += A[3 * j + i] * R[3 * i + j]; return r; };
  g[3] = ip(dphi); g[4] = ip(dth); g[5] = ip(dpsi);
}

// One job's GICP run: pcl's computeTransformation (:381-514) with its BFGS, as a coroutine.  Wherever the sequential code
// needs the device -- the correspondences of an outer iteration, a functor evaluation -- it posts the request and yields to
// the scheduler (gicp_run_tasks), which serves the requests of all live jobs together and resumes them with the answers.
struct GicpTask {
  smhip_context* h = nullptr;
  int job = 0, ns = 0;
  bool live = false;
  float guess[16];                    // base_transformation_ = guess (row-major float)
  float fin[16];                      // final_transformation_
  enum Req { kStart, kCorr, kFdf, kDone } req = kStart;
  double TRcm[16], R9[9];             // kCorr: transformation_ * guess (column-major) and its rotation (row-major)
  uint32_t ncorr = 0;                 // ... answered with the kept correspondences
  GicpPose P;                         // kFdf: the pose; answered in the job's row of out_pinned
  smhip_status status = SMHIP_OK;
  int evals = 0, it = 0;
  ucontext_t ctx, sched;
  std::unique_ptr<char[]> stack;
  void yield() { swapcontext(&ctx, &sched); }
  void body();
};

// One evaluation of the functor (f and g together: one launch either way), :250-377
struct GicpFunctor {
  GicpTask* t;
  void fdf(const double* x, double& f, double* g) {
    float T[16];
    apply_state_f32(t->guess, x, T);
    for (int i = 0; i < 12; ++i) { t->P.T[i] = T[i]; t->P.B[i] = t->guess[i]; }
    if (t->status == SMHIP_OK) { t->req = GicpTask::kFdf; t->yield(); }
    if (t->status) { f = 0; for (int i = 0; i < 6; ++i) g[i] = 0; return; }     // (the run is being abandoned: any finite answer ends it)
    const double* o = gicp_of(t->h).out_pinned + (size_t)t->job * kGicpCols;
    const double m = o[13];
    f = o[0] / m;
    for (int i = 0; i < 3; ++i) g[i] = o[1 + i] * (2.0 / m);
    double R[9];
    for (int i = 0; i < 9; ++i) R[i] = o[4 + i] * (2.0 / m);
    r_derivative(x, R, g);
    t->evals++;
  }
};

// ---- pcl/registration/bfgs.h: BFGS<Functor> (GSL vector_bfgs2, Fletcher's line search) -------------
enum { kBfgsSuccess = 0, kBfgsNoProgress = 1, kBfgsRunning = -1 };
constexpr int kGicpLazyMinJobs = 4;          // batches from this size on estimate target covariances on demand

int solve_quadratic(double a, double b, double c, double* x0, double* x1) {     // gsl_poly_solve_quadratic
  if (a == 0) { if (b == 0) return 0; *x0 = -c / b; return 1; }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) { const double r = std::sqrt(-c / a); *x0 = -r; *x1 = r; }
    else {
      const double sgnb = b > 0 ? 1 : -1;
      const double temp = -0.5 * (b + sgnb * std::sqrt(disc));
      const double r1 = temp / a, r2 = c / temp;
      if (r1 < r2) { *x0 = r1; *x1 = r2; } else { *x0 = r2; *x1 = r1; }
    }
    return 2;
  }
  if (disc == 0) { *x0 = -0.5 * b / a; *x1 = -0.5 * b / a; return 2; }
  return 0;
}
double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
  const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
  const double c = 2 * (f1 - f0 - fp0);
  double zmin = zl, fmin = fl;
  if (fh < fmin) { zmin = zh; fmin = fh; }
  if (c > 0) {
    const double z = -fp0 / c;
    if (z > zl && z < zh) { const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0)); if (f < fmin) { zmin = z; fmin = f; } }
  }
  return zmin;
}
double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1, xi = fp0 + fp1 - 2 * (f1 - f0);
  const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
  auto cubic = [&](double z) { return c0 + z * (c1 + z * (c2 + z * c3)); };
  double zmin = zl, fmin = cubic(zl);
  auto check = [&](double z) { const double y = cubic(z); if (y < fmin) { zmin = z; fmin = y; } };
  check(zh);
  double z0 = 0, z1 = 0;
  const int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
  if (n == 2) { if (z0 > zl && z0 < zh) check(z0); if (z1 > zl && z1 < zh) check(z1); }
  else if (n == 1) { if (z0 > zl && z0 < zh) check(z0); }
  return zmin;
}
double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
  double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a);
  if (zmin > zmax) std::swap(zmin, zmax);
  const double z = (order > 2 && std::isfinite(fpb)) ? interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax)
                                                      : interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
  return a + z * (b - a);
}

struct Bfgs {
  GicpFunctor& fn;
  double sigma = 0.01, rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;   // :218-223
  int order = 3;
  double x0[6], g0[6], p[6], gradient[6], f = 0, delta_f = 0, g0norm = 0, pnorm = 0, fp0 = 0;
  // the line function's one-entry caches (bfgs.h moveTo / applyF / applyDF / applyFDF)
  double x_alpha[6], g_alpha[6], f_alpha = 0, df_alpha = 0, x_key = 0, f_key = 0, g_key = 0, df_key = 0;
  explicit Bfgs(GicpFunctor& f_) : fn(f_) {}
  static double dot(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }
  static double norm(const double* a) { return std::sqrt(dot(a, a)); }
  void move(double alpha) { if (alpha != x_key) { for (int i = 0; i < 6; ++i) x_alpha[i] = x0[i] + alpha * p[i]; x_key = alpha; } }
  // the functor's f-only and df-only entry points cost the same launch as fdf here, so every miss fills both caches
  void eval(double alpha) {
    move(alpha);
    fn.fdf(x_alpha, f_alpha, g_alpha);
    f_key = g_key = alpha;
    df_alpha = dot(g_alpha, p); df_key = alpha;
  }
  double F(double alpha) { if (alpha != f_key) eval(alpha); return f_alpha; }
  double DF(double alpha) {
    if (alpha == df_key) return df_alpha;
    if (alpha != g_key) eval(alpha);
    else { df_alpha = dot(g_alpha, p); df_key = alpha; }
    return df_alpha;
  }
  void init(const double* x) {
    delta_f = 0;
    fn.fdf(x, f, gradient);
    for (int i = 0; i < 6; ++i) { x0[i] = x[i]; g0[i] = gradient[i]; }
    g0norm = norm(g0);
    for (int i = 0; i < 6; ++i) p[i] = -gradient[i] / g0norm;
    pnorm = norm(p);
    fp0 = -g0norm;
    for (int i = 0; i < 6; ++i) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }
    x_key = f_key = g_key = df_key = 0; f_alpha = f; df_alpha = dot(g_alpha, p);
  }
  int line_search(double alpha1, double* alpha_new) {
    const double f0v = F(0.0), fp0v = DF(0.0);
    double falpha, falpha_prev = f0v, fpalpha, fpalpha_prev = fp0v, alpha = alpha1, alpha_prev = 0;
    double a = 0, b = alpha, fa = f0v, fb = 0, fpa = fp0v, fpb = 0;
    int i = 0;
    bool bracketed = false;
    const bool dbg = std::getenv("SMHIP_GICP_DEBUG") != nullptr;
    if (dbg) std::fprintf(stderr, "[gicp]     ls f0=%.15g fp0=%.9g alpha1=%.9g\n", f0v, fp0v, alpha1);
    while (i++ < 100) {
      falpha = F(alpha);
      if (dbg) std::fprintf(stderr, "[gicp]     br alpha=%.12g f=%.15g\n", alpha, falpha);
      if (falpha > f0v + alpha * rho * fp0v || falpha >= falpha_prev) {
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev; b = alpha; fb = falpha; fpb = NAN; bracketed = true; break;
      }
      fpalpha = DF(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0v) { *alpha_new = alpha; return kBfgsSuccess; }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha; b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev; bracketed = true; break;
      }
      const double delta = alpha - alpha_prev;
      const double alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
    }
    (void)bracketed;
    while (i++ < 100) {
      const double delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
      falpha = F(alpha);
      if (dbg && i < 12) std::fprintf(stderr, "[gicp]     sec a=%.9g b=%.9g alpha=%.12g f=%.15g fa=%.15g fpa=%.6g\n", a, b, alpha, falpha, fa, fpa);
      if ((a - alpha) * fpa <= 2.220446049250313e-16) return kBfgsNoProgress;
      if (falpha > f0v + rho * alpha * fp0v || falpha >= fa) { b = alpha; fb = falpha; fpb = NAN; }
      else {
        fpalpha = DF(alpha);
        if (std::fabs(fpalpha) <= -sigma * fp0v) { *alpha_new = alpha; return kBfgsSuccess; }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; fb = fa; fpb = fpa; }
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
    *alpha_new = alpha;
    return kBfgsSuccess;
  }
  int one_step(double* x) {
    const double f0v = f;
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) return kBfgsNoProgress;
    double alpha1;
    if (delta_f < 0) { const double del = std::max(-delta_f, 10 * 2.220446049250313e-16 * std::fabs(f0v)); alpha1 = std::min(1.0, 2.0 * del / (-fp0)); }
    else alpha1 = std::fabs(step_size);
    double alpha = 0;
    const int st = line_search(alpha1, &alpha);
    if (st != kBfgsSuccess) return st;
    f = F(alpha); (void)DF(alpha);                               // updatePosition
    for (int i = 0; i < 6; ++i) { x[i] = x_alpha[i]; gradient[i] = g_alpha[i]; }
    delta_f = f - f0v;
    double dx0[6], dg0[6];
    for (int i = 0; i < 6; ++i) { dx0[i] = x[i] - x0[i]; dg0[i] = gradient[i] - g0[i]; }
    const double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0);
    double A = 0, B = 0;
    if (dxdg != 0) { B = dxg / dxdg; A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg; }
    for (int i = 0; i < 6; ++i) p[i] = gradient[i] - A * dx0[i] - B * dg0[i];
    for (int i = 0; i < 6; ++i) { g0[i] = gradient[i]; x0[i] = x[i]; }
    g0norm = norm(g0);
    pnorm = norm(p);
    const double dir = dot(p, gradient) > 0 ? -1.0 : 1.0;
    for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    for (int i = 0; i < 6; ++i) { x_alpha[i] = x0[i]; g_alpha[i] = g0[i]; }   // changeDirection
    x_key = f_key = g_key = df_key = 0; df_alpha = dot(g_alpha, p);
    return kBfgsSuccess;
  }
  int test_gradient(double eps) const { return g0norm < eps ? kBfgsSuccess : kBfgsRunning; }
};

// ring-search settings of the GICP searches (exact distances, no tie-order requirement, no previous match to seed a ball)
struct GicpSearchMode {
  smhip_context* h; int sort_was, ball_was, ring_was; float cutoff_was;
  explicit GicpSearchMode(smhip_context* h_) : h(h_), sort_was(h_->dev.sort_cells), ball_was(h_->dev.use_ball), ring_was(h_->dev.max_ring), cutoff_was(h_->dev.nn_cutoff2) {
    h->dev.sort_cells = 0; h->dev.use_ball = 0;
  }
  ~GicpSearchMode() { h->dev.sort_cells = sort_was; h->dev.use_ball = ball_was; h->dev.max_ring = ring_was; h->dev.nn_cutoff2 = cutoff_was; }
};

// Which grid cell an Align searches with.  The searches are exact whatever the cell; what differs is their cost.  A 20-NN
// neighbourhood wants cells of a few point spacings (0.6 m: one shell holds the 20 neighbours almost everywhere; on the handle's
// small cell it walks hundreds of grid rows), the 1-NN searches (NDT fitness, correspondences, final fitness) are a third faster
// on the small cell.  An Align that may have to estimate target covariances builds the target's grid ONCE, with the k-NN cell,
// and runs its 1-NN searches on it too; an Align whose targets all hold a complete, current set of covariances stays on the
// small cell (that grid is kept as well).  The scratch slots (a source as its own target) always get the k-NN cell.
inline float gicp_knn_cell(const smhip_ndt_gicp_options& o) {
  return o.gicp_search_cell > 0 ? o.gicp_search_cell : 3.0f * (o.using_voxel_filter ? o.voxel_resolution : 0.2f);
}
struct GicpCell {
  smhip_context* h; float was;
  GicpCell(smhip_context* h_, float cell) : h(h_), was(h_->dev.grid_cell) { h->dev.grid_cell = cell; }
  ~GicpCell() { h->dev.grid_cell = was; }
};

// pair inputs of slots [first, first + K) (guesses: column-major 4x4 each); the GICP clouds carry no normals
smhip_status gicp_fill_inputs(smhip_context* h, int first, int K, const double* guesses, int* ns_max, int* nt_max) {
  std::vector<int> had((size_t)K);
  for (int k = 0; k < K; ++k) { had[k] = h->has_normals[first + k]; h->has_normals[first + k] = 1; }
  const smhip_status s = fill_inputs(h, K, guesses, ns_max, nt_max, first);
  for (int k = 0; k < K; ++k) h->has_normals[first + k] = had[k];
  return s;
}

// search structures over the targets of slots [first, first + K), built (all of them, one pass) unless every one is current
smhip_status gicp_build_grids(smhip_context* h, int first, int K, bool* kept) {
  bool cached = true;
  for (int p = first; p < first + K; ++p) cached = cached && grid_cached(h, p);
  *kept = cached;
  if (cached) return SMHIP_OK;
  std::vector<double> I((size_t)16 * K, 0.0);
  for (int k = 0; k < K; ++k) for (int i = 0; i < 4; ++i) I[(size_t)16 * k + 5 * i] = 1.0;
  int ns_max = 0, nt_max = 0;
  smhip_status s = gicp_fill_inputs(h, first, K, I.data(), &ns_max, &nt_max);
  if (s) return s;
  s = enqueue_resets(h, K, first);
  if (s) return s;
  return enqueue_grid_build(h, whole_batch(h, K, first), nt_max);
}

// One round of the batch: the correspondence steps (:405-463: 1-NN of the moved source, distance filter, Mahalanobis matrices)
// of the jobs in `corr` and the functor evaluations of the jobs in `fdf`, all jobs of slots [first, first + K)
smhip_status gicp_round(smhip_context* h, int first, int K, const std::vector<GicpTask*>& corr, const std::vector<GicpTask*>& fdf) {
  GicpHost& G = gicp_of(h);
  const smhip_ndt_gicp_options& o = G.opts;
  if (!corr.empty()) {
    // pair inputs: the jobs of this step get their pose, the others no queries (their slots are skipped by every launch)
    std::vector<double> g((size_t)16 * K, 0.0);
    for (int k = 0; k < K; ++k) for (int i = 0; i < 4; ++i) g[(size_t)16 * k + 5 * i] = 1.0;
    std::vector<char> in_step((size_t)K, 0);
    for (const GicpTask* t : corr) { std::memcpy(&g[(size_t)16 * (t->job - first)], t->TRcm, sizeof(double) * 16); in_step[t->job - first] = 1; }
    int ns_max = 0, nt_max = 0;
    smhip_status s = gicp_fill_inputs(h, first, K, g.data(), &ns_max, &nt_max);
    if (s) return s;
    ns_max = 0;
    for (int k = 0; k < K; ++k) {
      if (in_step[k]) ns_max = std::max(ns_max, h->ns[first + k]);
      else h->in_pinned[first + k].ns = 0;
    }
    const float thr2 = (float)(o.gicp_corr_dist_threshold * o.gicp_corr_dist_threshold);
    {
      GicpSearchMode mode(h);
      bool cached = true;
      for (int p = first; p < first + K; ++p) cached = cached && grid_cached(h, p);
      const Half f = whole_batch(h, K, first);
      if (cached) {
        HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(f.d.in) + first, h->in_pinned + first, sizeof(PairInput) * K, hipMemcpyHostToDevice, h->stream));
        s = ensure_packed(h, first, K);
        if (s) return s;
        hipLaunchKernelGGL(reset_scratch_light, dim3(std::min(1024, 8 * K)), dim3(256), 0, h->stream, f.d, first, K);
        hipLaunchKernelGGL(pose_setup, dim3(ceil_div(K, 64)), dim3(64), 0, h->stream, f.d, K);
        h->cache_hits++;
      } else {
        s = enqueue_resets(h, K, first);
        if (s == SMHIP_OK) s = enqueue_grid_build(h, f, nt_max);
        if (s) return s;
      }
      h->dev.nn_cutoff2 = thr2;          // correspondences beyond the distance threshold are dropped anyway (gicp_corr)
      // with the row-occupancy bitmap a wide ring is cheap: let the ring search reach the correspondence distance (5 m)
      // instead of handing the far queries to the brute-force sweep over the whole 0.5 M-point target
      h->dev.max_ring = std::max(h->dev.max_ring, 32);
      s = enqueue_find_closests_half(h, whole_batch(h, K, first), ns_max, 0);
      if (s) return s;
    }
    HIPCHK(h, hipMemsetAsync(h->dev.hist + (size_t)first * kHistBins, 0, sizeof(uint32_t) * kHistBins * (size_t)K, h->stream));
    h->ev_used = 0;
    HIPCHK(h, hipMemsetAsync(G.dev.count + first, 0, sizeof(uint32_t) * (size_t)K, h->stream));
    HIPCHK(h, hipMemsetAsync(G.dev.need_count + first, 0, sizeof(uint32_t) * (size_t)K, h->stream));
    // the covariances of the matched target points that do not have one yet
    std::vector<const GicpTask*> lazy;
    for (const GicpTask* t : corr) if (!G.job[t->job].cov_full) lazy.push_back(t);
    for (size_t c0 = 0; c0 < lazy.size(); c0 += kGicpLaunchJobs) {
      GicpNeedBatch N{};
      N.n = (int)std::min<size_t>(kGicpLaunchJobs, lazy.size() - c0);
      N.k = o.gicp_k_correspondences; N.gicp_epsilon = o.gicp_epsilon;
      int nmax = 0;
      for (int e = 0; e < N.n; ++e) {
        const GicpTask* t = lazy[c0 + e];
        N.j[e].job = t->job; N.j[e].ns = t->ns; N.j[e].thr2 = thr2; N.j[e].epoch = G.job[t->job].epoch;
        nmax = std::max(nmax, t->ns);
      }
      IcpDev dk = h->dev;
      dk.have_rowbits = 1;                     // the job slots' grids are ring-search structures
      hipLaunchKernelGGL(gicp_need, dim3(ceil_div(nmax, 256), N.n), dim3(256), 0, h->stream, h->dev, G.dev, N);
      if (N.k <= 20) hipLaunchKernelGGL(gicp_knn_cov_listed<20>, dim3(ceil_div(nmax, kGicpKnnThreads), N.n), dim3(kGicpKnnThreads), 0, h->stream, dk, G.dev, N);
      else hipLaunchKernelGGL(gicp_knn_cov_listed<kGicpKMax>, dim3(ceil_div(nmax, kGicpKnnThreads), N.n), dim3(kGicpKnnThreads), 0, h->stream, dk, G.dev, N);
    }
    for (size_t c0 = 0; c0 < corr.size(); c0 += kGicpLaunchJobs) {
      GicpCorrBatch L{};
      L.n = (int)std::min<size_t>(kGicpLaunchJobs, corr.size() - c0);
      int nmax = 0;
      for (int e = 0; e < L.n; ++e) {
        const GicpTask* t = corr[c0 + e];
        L.j[e].job = t->job; L.j[e].ns = t->ns; L.j[e].thr2 = thr2;
        for (int i = 0; i < 9; ++i) L.j[e].R[i] = t->R9[i];
        nmax = std::max(nmax, t->ns);
      }
      hipLaunchKernelGGL(gicp_corr, dim3(ceil_div(nmax, 256), L.n), dim3(256), 0, h->stream, h->dev, G.dev, L);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(G.count_pinned + first, G.dev.count + first, sizeof(uint32_t) * (size_t)K, hipMemcpyDeviceToHost, h->stream));
  }
  unsigned long long seq = 0;
  if (!fdf.empty()) {
    seq = ++G.seq;
    for (size_t c0 = 0; c0 < fdf.size(); c0 += kGicpLaunchJobs) {
      GicpFdfBatch L{};
      L.n = (int)std::min<size_t>(kGicpLaunchJobs, fdf.size() - c0);
      L.total = (int)fdf.size();
      L.seq = seq;
      int bmax = 1;
      for (int e = 0; e < L.n; ++e) {
        const GicpTask* t = fdf[c0 + e];
        L.j[e].job = t->job; L.j[e].ns = t->ns;
        L.j[e].nblk = std::min(kGicpMaxBlocks, std::max(1, ceil_div(t->ns, 256)));
        L.j[e].P = t->P;
        bmax = std::max(bmax, L.j[e].nblk);
      }
      // its last workgroup stores the sums and, after the round's last job, the round's number into page-locked memory
      hipLaunchKernelGGL(gicp_fdf, dim3(bmax, L.n), dim3(256), 0, h->stream, h->dev, G.dev, L);
    }
  }
  bool ok = hipGetLastError() == hipSuccess;
  if (ok && !corr.empty()) {
    ok = hipStreamSynchronize(h->stream) == hipSuccess;
  } else if (ok) {
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(G.out_pinned + (size_t)G.jobs * kGicpCols);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0x3ffu) == 0) {
        if (spins > (1u << 16)) std::this_thread::yield();                     // a long evaluation (shared GPU): stop burning the core
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
          // slow is not failed (an oversubscribed GPU, six pooled matchers): wait for the stream the ordinary way and look again
          ok = hipStreamSynchronize(h->stream) == hipSuccess && hipGetLastError() == hipSuccess &&
               __atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) == seq;
          break;
        }
      }
    }
  }
  if (!ok) {
    // a launch was lost or died before its last workgroup: leave the counters clean for the next evaluation
    (void)hipStreamSynchronize(h->stream);
    (void)hipMemsetAsync(G.dev.ticket, 0, sizeof(uint32_t) * (size_t)G.jobs, h->stream);
    (void)hipMemsetAsync(G.dev.round_done, 0, sizeof(uint32_t), h->stream);
    h->err = "GICP: a correspondence / functor launch failed";
    return SMHIP_ERR_HIP;
  }
  for (GicpTask* t : corr) t->ncorr = G.count_pinned[t->job];
  return SMHIP_OK;
}

// pcl GICP computeTransformation's outer loop (:404-514) on the job's clouds; covariances are in place
void GicpTask::body() {
  const smhip_ndt_gicp_options& o = gicp_of(h).opts;
  float T[16], prev[16];                                   // transformation_, previous_transformation_
  for (int i = 0; i < 16; ++i) T[i] = prev[i] = (i % 5 == 0) ? 1.f : 0.f;
  GicpFunctor fn{this};
  it = 0; evals = 0; ncorr = 0;
  const bool dbg = std::getenv("SMHIP_GICP_DEBUG") != nullptr;
  for (;;) {
    // transform_R = transformation_ * guess in double (:425-429); the search uses the same product
    double TR[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double a = 0; for (int q = 0; q < 4; ++q) a += (double)T[4 * i + q] * (double)guess[4 * q + j]; TR[4 * i + j] = a; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) TRcm[4 * j + i] = TR[4 * i + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R9[3 * i + j] = TR[4 * i + j];
    req = kCorr; yield();
    if (status) return;
    for (int i = 0; i < 16; ++i) prev[i] = T[i];                          // :467
    if (ncorr < 4) break;                                                // NotEnoughPointsException -> break (:494-498)
    // ---- estimateRigidTransformationBFGS (:189-247)
    double x[6] = {T[3], T[7], T[11], std::atan2((double)T[9], (double)T[10]), std::asin(-(double)T[8]), std::atan2((double)T[4], (double)T[0])};
    Bfgs bfgs(fn);
    bfgs.init(x);
    int inner = 0, result;
    if (dbg) std::fprintf(stderr, "[gicp] job %d outer %d m=%u f0=%.12g g0=(%.6g %.6g %.6g %.6g %.6g %.6g)\n", job, it, ncorr, bfgs.f, bfgs.g0[0], bfgs.g0[1], bfgs.g0[2], bfgs.g0[3], bfgs.g0[4], bfgs.g0[5]);
    do {
      inner++;
      result = bfgs.one_step(x);
      if (dbg) std::fprintf(stderr, "[gicp]   inner %d status %d f=%.12g |g|=%.6g x=(%.8g %.8g %.8g %.8g %.8g %.8g) evals=%d\n", inner, result, bfgs.f, bfgs.g0norm, x[0], x[1], x[2], x[3], x[4], x[5], evals);
      if (result) break;
      result = bfgs.test_gradient(1e-2);
    } while (result == kBfgsRunning && inner < o.gicp_max_inner_iterations && status == SMHIP_OK);
    if (status) return;
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    apply_state_f32(I4, x, T);                                           // :240-241
    double delta = 0;                                                    // :475-491
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        const double ratio = (r < 3 && c < 3) ? 1.0 / o.gicp_rotation_epsilon : 1.0 / o.gicp_transformation_epsilon;
        delta = std::max(delta, ratio * std::fabs((double)prev[4 * r + c] - (double)T[4 * r + c]));
      }
    it++;
    if (it >= o.gicp_max_iterations || delta < 1) { for (int i = 0; i < 16; ++i) prev[i] = T[i]; break; }   // :500-505
  }
  // final_transformation_ (:511-514)
  for (int i = 0; i < 16; ++i) fin[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) { float a = 0; for (int q = 0; q < 3; ++q) a += prev[4 * r + q] * guess[4 * q + c]; fin[4 * r + c] = a; }
    fin[4 * r + 3] = prev[4 * r + 3] + guess[4 * r + 3];
  }
}

void gicp_task_entry(unsigned lo, unsigned hi) {
  GicpTask* t = reinterpret_cast<GicpTask*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
  t->body();
  t->req = GicpTask::kDone;
  t->yield();                                              // never resumed
}

// the live tasks (jobs of slots [first, first + K)) advanced in lock-step until each has finished
smhip_status gicp_run_tasks(smhip_context* h, int first, int K, std::vector<GicpTask>& tasks) {
  constexpr size_t kStack = 256 * 1024;
  for (GicpTask& t : tasks) {
    if (!t.live) continue;
    t.stack.reset(new char[kStack]);
    if (getcontext(&t.ctx) != 0) { h->err = "getcontext failed"; return SMHIP_ERR_HIP; }
    t.ctx.uc_stack.ss_sp = t.stack.get();
    t.ctx.uc_stack.ss_size = kStack;
    t.ctx.uc_link = nullptr;
    const uintptr_t p = reinterpret_cast<uintptr_t>(&t);
    makecontext(&t.ctx, reinterpret_cast<void (*)()>(gicp_task_entry), 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
  }
  std::vector<GicpTask*> corr, fdf;
  for (;;) {
    corr.clear(); fdf.clear();
    for (GicpTask& t : tasks) {
      if (!t.live) continue;
      swapcontext(&t.sched, &t.ctx);                       // runs until the job's next request
      if (t.req == GicpTask::kDone) t.live = false;
      else (t.req == GicpTask::kCorr ? corr : fdf).push_back(&t);
    }
    if (corr.empty() && fdf.empty()) return SMHIP_OK;
    const smhip_status s = gicp_round(h, first, K, corr, fdf);
    if (s) return s;                                       // (the unfinished coroutines are dropped with their stacks)
  }
}

// GICP of the jobs first .. first + K - 1 whose `run` flag is set: covariances (:391-402), then the lock-step outer loops.
// guess / fin: float 4x4 row-major per job
smhip_status gicp_align_jobs(smhip_context* h, int first, int K, const char* run, const float* guess, float* fin, smhip_ndt_gicp_stats* stats) {
  GicpHost& G = gicp_of(h);
  const smhip_ndt_gicp_options& o = G.opts;
  const int k = o.gicp_k_correspondences;
  if (k < 3 || k > kGicpKMax) { h->err = "gicp_k_correspondences must be in [3, 32]"; return SMHIP_ERR_INVALID_ARGUMENT; }
  const int sfirst = gicp_scratch_slot(h, first);
  for (int e = 0; e < K; ++e) {
    const int ns = h->ns[first + e], nt = h->nt[first + e];
    if (ns < k || nt < k) { h->err = "GICP: a cloud has fewer points than k_correspondences (gicp_omp_impl.hpp:64-68)"; return SMHIP_ERR_INVALID_ARGUMENT; }
    if (ns > h->dev.nt_cap) { h->err = "GICP: max_target_points must be >= the down-sampled source size"; return SMHIP_ERR_CAPACITY; }
  }
  // ---- covariances: a target's over its job slot's grid, a source's over the scratch slot's (the source as its own target)
  for (int e = 0; e < K; ++e) {
    const int ns = h->ns[first + e];
    hipLaunchKernelGGL(gicp_copy_points, dim3(ceil_div(ns, 256)), dim3(256), 0, h->stream, h->dev.src + (size_t)(first + e) * h->dev.ns_cap,
                       const_cast<float4*>(h->dev.tgt_p) + (size_t)(sfirst + e) * h->dev.nt_cap, ns);
    h->ns[sfirst + e] = ns; h->nt[sfirst + e] = ns;
    touch_target(h, sfirst + e);
  }
  {
    GicpSearchMode mode(h);
    bool kept = false;
    smhip_status s;
    { GicpCell scratch_cell(h, gicp_knn_cell(o)); s = gicp_build_grids(h, sfirst, K, &kept); }
    if (s == SMHIP_OK) s = gicp_build_grids(h, first, K, &kept);
    if (s) return s;
    if (kept) h->cache_hits++;
  }
  {
    IcpDev dk = h->dev;
    dk.have_rowbits = 1;                       // built by gicp_build_grids (a ring-search context)
    GicpKnnBatch L{};
    int nmax = 0;
    auto flush = [&]() {
      if (!L.n) return;
      // the set's LDS footprint follows k (the default 20 fits 20 entries per thread: 20 KiB per workgroup instead of 32)
      if (k <= 20) hipLaunchKernelGGL(gicp_knn_cov<20>, dim3(ceil_div(nmax, kGicpKnnThreads), L.n), dim3(kGicpKnnThreads), 0, h->stream, dk, L, k, o.gicp_epsilon);
      else hipLaunchKernelGGL(gicp_knn_cov<kGicpKMax>, dim3(ceil_div(nmax, kGicpKnnThreads), L.n), dim3(kGicpKnnThreads), 0, h->stream, dk, L, k, o.gicp_epsilon);
      L.n = 0; nmax = 0;
    };
    // Target covariances.  A batch estimates them on demand (gicp_need / gicp_knn_cov_listed in the correspondence steps: a
    // twelfth of the neighbourhoods); a single Align estimates them all up front, as computeCovariances does (:391-402): the
    // duration of a neighbourhood launch is that of its slowest queries (far-range points with metres to their 20th neighbour),
    // which a launch over 574 000 points hides no worse than one over 34 000, and a complete set lets the following Aligns on
    // the kept target skip the step altogether.
    const bool eager = K < kGicpLazyMinJobs;
    for (int e = 0; e < K; ++e) {
      if (!run[e]) continue;
      GicpJobHost& jh = G.job[first + e];
      // a target's covariances depend on the target, k and epsilon alone: the entries estimated so far stay while those do
      const bool cov_kept = h->target_cache && jh.epoch != 0 && jh.cov_gen == h->tgt_gen[first + e] && jh.cov_k == k && jh.cov_eps == o.gicp_epsilon;
      if (cov_kept) {
        h->cache_hits++;
      } else {
        jh.cov_full = eager;
        if (eager) {
          L.slot[L.n] = first + e; L.cov[L.n] = G.dev.cov_t + (size_t)(first + e) * h->dev.nt_cap * 6; ++L.n;
          nmax = std::max(nmax, h->nt[first + e]);
          if (L.n == kGicpKnnJobs) flush();
        }
        if (jh.epoch == 0xffffffffu) {         // (4 billion invalidations later: start over)
          HIPCHK(h, hipMemsetAsync(G.dev.cov_epoch + (size_t)(first + e) * h->dev.nt_cap, 0, sizeof(uint32_t) * (size_t)h->dev.nt_cap, h->stream));
          jh.epoch = 0;
        }
        ++jh.epoch;
        jh.cov_gen = h->tgt_gen[first + e]; jh.cov_k = k; jh.cov_eps = o.gicp_epsilon;
      }
      // the source's covariances: all of them, every Align (every source point is a query)
      L.slot[L.n] = sfirst + e; L.cov[L.n] = G.dev.cov_s + (size_t)(first + e) * h->dev.ns_cap * 6; ++L.n;
      nmax = std::max(nmax, h->ns[first + e]);
      if (L.n == kGicpKnnJobs) flush();
    }
    flush();
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));              // the pinned pair inputs are rewritten below
  // ---- outer loops
  std::vector<GicpTask> tasks((size_t)K);
  for (int e = 0; e < K; ++e) {
    GicpTask& t = tasks[e];
    t.h = h; t.job = first + e; t.ns = h->ns[first + e]; t.live = run[e] != 0;
    for (int i = 0; i < 16; ++i) t.guess[i] = guess[16 * e + i];
  }
  const smhip_status s = gicp_run_tasks(h, first, K, tasks);
  if (s) return s;
  for (int e = 0; e < K; ++e) {
    if (!run[e]) continue;
    const GicpTask& t = tasks[e];
    for (int i = 0; i < 16; ++i) fin[16 * e + i] = t.fin[i];
    if (stats) { stats[e].gicp_iterations = t.it; stats[e].gicp_function_evaluations = t.evals; stats[e].gicp_correspondences = (int32_t)t.ncorr; }
    G.evals = t.evals;
  }
  return SMHIP_OK;
}

smhip_status gicp_upload_raw(smhip_context* h, float4* dst, const float* xyz, int stride, int n, int cap, int* n_out) {
  if (!xyz || n <= 0 || (stride != 3 && stride != 4 && stride != 5)) { h->err = "bad cloud / stride"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > cap) { h->err = "cloud larger than the handle's capacity"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // the pinned staging buffer holds 2 * max(ns_cap, nt_cap) points
  for (int i = 0; i < n; ++i) h->stage[i] = make_float4(xyz[(size_t)stride * i], xyz[(size_t)stride * i + 1], xyz[(size_t)stride * i + 2], 0.f);
  HIPCHK(h, hipMemcpyAsync(dst, h->stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *n_out = n;
  return SMHIP_OK;
}

void colmajor_to_rm_f32(const double* cm, float* rm) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) rm[4 * r + c] = (float)cm[4 * c + r]; }
void rm_f32_to_colmajor(const float* rm, double* cm) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) cm[4 * c + r] = (double)rm[4 * r + c]; }

}  // namespace

extern "C" {

void smhip_ndt_gicp_default_options(smhip_ndt_gicp_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->voxel_resolution = 0.2f;              // ndt_gicp.h:73
  o->using_voxel_filter = 1;               // :74
  o->use_ndt = 1;                          // :75
  o->ndt_transformation_epsilon = 0.01f;   // ndt_gicp.cc:38
  o->ndt_step_size = 0.1f;                 // :39
  o->ndt_resolution = 1.0f;                // :40
  o->ndt_max_iterations = 35;              // :41
  o->gicp_rotation_epsilon = 1e-3;         // :43
  o->gicp_max_iterations = 35;             // :44
  o->gicp_transformation_epsilon = 5e-4;   // PCL defaults, gicp_omp.h:108-118
  o->gicp_epsilon = 1e-3;
  o->gicp_corr_dist_threshold = 5.0;
  o->gicp_max_inner_iterations = 20;
  o->gicp_k_correspondences = 20;
}

smhip_status smhip_ndt_gicp_set_options(smhip_handle h, const smhip_ndt_gicp_options* o) {
  if (!h || !o) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!(o->voxel_resolution > 0) || !(o->ndt_resolution > 0) || !(o->ndt_step_size > 0) || o->gicp_max_iterations < 1 ||
      o->gicp_k_correspondences < 3 || o->gicp_k_correspondences > kGicpKMax || !(o->gicp_rotation_epsilon > 0) ||
      !(o->gicp_transformation_epsilon > 0)) {
    h->err = "bad NdtWithGicp options";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  gicp_of(h).opts = *o;
  return SMHIP_OK;
}

static smhip_status gicp_check_job(smhip_handle h, int job) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  smhip_status s = gicp_ensure(h);
  if (s) return s;
  if (job < 0 || job >= gicp_of(h).jobs) { h->err = "NdtWithGicp job out of range (a handle runs pair_slots / 2 jobs)"; return SMHIP_ERR_INVALID_ARGUMENT; }
  return SMHIP_OK;
}

int smhip_ndt_gicp_jobs(smhip_handle h) { return h ? h->dev.slots / 2 : 0; }

smhip_status smhip_ndt_gicp_set_source_f32_job(smhip_handle h, int job, const float* xyz, int stride_floats, int n) {
  smhip_status s = gicp_check_job(h, job);
  if (s) return s;
  GicpHost& g = gicp_of(h);
  return gicp_upload_raw(h, g.raw_src + (size_t)job * h->dev.ns_cap, xyz, stride_floats, n, h->dev.ns_cap, &g.job[job].n_raw_src);
}

smhip_status smhip_ndt_gicp_set_target_f32_job(smhip_handle h, int job, const float* xyz, int stride_floats, int n) {
  smhip_status s = gicp_check_job(h, job);
  if (s) return s;
  GicpHost& g = gicp_of(h);
  g.job[job].raw_tgt_gen = ++h->gen_counter;
  return gicp_upload_raw(h, g.raw_tgt + (size_t)job * h->dev.nt_cap, xyz, stride_floats, n, h->dev.nt_cap, &g.job[job].n_raw_tgt);
}

smhip_status smhip_ndt_gicp_set_source_f32(smhip_handle h, const float* xyz, int stride_floats, int n) {
  return smhip_ndt_gicp_set_source_f32_job(h, 0, xyz, stride_floats, n);
}

smhip_status smhip_ndt_gicp_set_target_f32(smhip_handle h, const float* xyz, int stride_floats, int n) {
  return smhip_ndt_gicp_set_target_f32_job(h, 0, xyz, stride_floats, n);
}

// ndt_gicp.cc:55-81: the (optionally down-sampled) clouds become the job slot's source and target
static smhip_status ndt_gicp_stage_clouds(smhip_handle h, int job) {
  GicpHost& g = gicp_of(h);
  GicpJobHost& jh = g.job[job];
  if (jh.n_raw_src <= 0 || jh.n_raw_tgt <= 0) { h->err = "NdtWithGicp::Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
  float4* src0 = const_cast<float4*>(h->dev.src) + (size_t)job * h->dev.ns_cap;
  float4* tgt0 = const_cast<float4*>(h->dev.tgt_p) + (size_t)job * h->dev.nt_cap;
  const float4* raw_src = g.raw_src + (size_t)job * h->dev.ns_cap;
  const float4* raw_tgt = g.raw_tgt + (size_t)job * h->dev.nt_cap;
  int ms = jh.n_raw_src, mt = jh.n_raw_tgt;
  hipError_t e = hipSuccess;
  const int filt = g.opts.using_voxel_filter ? 1 : 0;
  const bool keep_target = h->target_cache && jh.raw_tgt_gen != 0 && jh.staged_raw_gen == jh.raw_tgt_gen && jh.staged_slot_gen == h->tgt_gen[job] &&
                           jh.staged_filter == filt && (!filt || jh.staged_res == g.opts.voxel_resolution);
  if (filt) {
    e = prep_approx_voxel_grid(h->prep, h->stream, raw_src, jh.n_raw_src, g.opts.voxel_resolution, g.ds_tmp, &ms);
    if (e == hipSuccess) e = prep_morton_sort(h->prep, h->stream, g.ds_tmp, ms, src0);
    if (e == hipSuccess && !keep_target) e = prep_approx_voxel_grid(h->prep, h->stream, raw_tgt, jh.n_raw_tgt, g.opts.voxel_resolution, tgt0, &mt);
  } else {
    e = prep_morton_sort(h->prep, h->stream, raw_src, ms, src0);
    if (e == hipSuccess && !keep_target) e = hipMemcpyAsync(tgt0, raw_tgt, sizeof(float4) * (size_t)mt, hipMemcpyDeviceToDevice, h->stream);
  }
  if (e != hipSuccess) { h->err = std::string("NdtWithGicp down-sampling: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  h->ns[job] = ms; h->has_normals[job] = 0;
  touch_source(h, job);
  if (keep_target) {
    h->cache_hits++;
    return SMHIP_OK;
  }
  h->nt[job] = mt;
  touch_target(h, job);
  jh.staged_raw_gen = jh.raw_tgt_gen; jh.staged_slot_gen = h->tgt_gen[job]; jh.staged_filter = filt; jh.staged_res = g.opts.voxel_resolution; jh.staged_nt = mt;
  return SMHIP_OK;
}

// the same for jobs first .. first + K - 1 at once when the voxel filter is on: ONE ApproximateVoxelGrid pass over every cloud that
// has to be filtered (all sources, and the targets that are not kept), one Morton ordering of the down-sampled sources, one
// synchronise -- instead of two sorts, a scan and a synchronise per cloud
static smhip_status ndt_gicp_stage_clouds_batch(smhip_handle h, int first, int K) {
  GicpHost& g = gicp_of(h);
  const size_t NS = h->dev.ns_cap, NT = h->dev.nt_cap;
  if (!g.prep_avg) {
    const size_t cap = (size_t)g.jobs * (NS + NT);
    if (cap > (size_t)0x7fffffff) { h->err = "NdtWithGicp: the handle's clouds together exceed the batched filter's 2^31 points"; return SMHIP_ERR_CAPACITY; }
    g.prep_avg = prep_create((int)cap);
    if (!g.prep_avg) { h->err = "NdtWithGicp: batched filter workspace allocation failed"; return SMHIP_ERR_HIP; }
  }
  if (!g.ds_src) { smhip_status s0 = dev_alloc(h, &g.ds_src, (size_t)g.jobs * NS); if (s0) return s0; }
  std::vector<const float4*> raw;
  std::vector<float4*> out;
  std::vector<int> n, owner;                        // owner: job e's source = 2 e, target = 2 e + 1
  std::vector<char> keep((size_t)K, 0);
  for (int e = 0; e < K; ++e) {
    const int job = first + e;
    GicpJobHost& jh = g.job[job];
    if (jh.n_raw_src <= 0 || jh.n_raw_tgt <= 0) { h->err = "NdtWithGicp::Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
    keep[e] = h->target_cache && jh.raw_tgt_gen != 0 && jh.staged_raw_gen == jh.raw_tgt_gen && jh.staged_slot_gen == h->tgt_gen[job] &&
              jh.staged_filter == 1 && jh.staged_res == g.opts.voxel_resolution;
    raw.push_back(g.raw_src + (size_t)job * NS); out.push_back(g.ds_src + (size_t)job * NS); n.push_back(jh.n_raw_src); owner.push_back(2 * e);
    if (!keep[e]) {
      raw.push_back(g.raw_tgt + (size_t)job * NT); out.push_back(const_cast<float4*>(h->dev.tgt_p) + (size_t)job * NT); n.push_back(jh.n_raw_tgt); owner.push_back(2 * e + 1);
    }
  }
  std::vector<int> m(raw.size());
  hipError_t er = prep_approx_voxel_grid_batch(g.prep_avg, h->stream, (int)raw.size(), raw.data(), n.data(), g.opts.voxel_resolution, out.data(), m.data());
  if (er != hipSuccess) { h->err = std::string("NdtWithGicp down-sampling: ") + hipGetErrorString(er); return SMHIP_ERR_HIP; }
  std::vector<int> stage_off((size_t)K), ms((size_t)K);
  std::vector<long long> out_off((size_t)K);
  for (size_t c = 0; c < raw.size(); ++c) {
    const int e = owner[c] >> 1, job = first + e;
    if (owner[c] & 1) {
      GicpJobHost& jh = g.job[job];
      h->nt[job] = m[c];
      touch_target(h, job);
      jh.staged_raw_gen = jh.raw_tgt_gen; jh.staged_slot_gen = h->tgt_gen[job]; jh.staged_filter = 1; jh.staged_res = g.opts.voxel_resolution; jh.staged_nt = m[c];
    } else {
      ms[e] = m[c]; stage_off[e] = (int)((size_t)job * NS); out_off[e] = (long long)((size_t)job * NS);
      h->ns[job] = m[c]; h->has_normals[job] = 0;
      touch_source(h, job);
    }
  }
  for (int e = 0; e < K; ++e) if (keep[e]) h->cache_hits++;
  er = prep_morton_sort_batch(g.prep_avg, h->stream, g.ds_src, K, stage_off.data(), ms.data(), out_off.data(), const_cast<float4*>(h->dev.src));
  if (er != hipSuccess) { h->err = std::string("NdtWithGicp source ordering: ") + hipGetErrorString(er); return SMHIP_ERR_HIP; }
  return SMHIP_OK;
}

// NdtWithGicp::Align of the jobs first .. first + K - 1, each stage over all of them at once
static smhip_status ndt_gicp_align_jobs(smhip_handle h, int first, int K, const double* guesses, double* results, double* scores, smhip_ndt_gicp_stats* stats) {
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = gicp_ensure(h);
  if (s) return s;
  GicpHost& g = gicp_of(h);
  if (first < 0 || K < 1 || first + K > g.jobs) { h->err = "NdtWithGicp jobs out of range (a handle runs pair_slots / 2 jobs)"; return SMHIP_ERR_INVALID_ARGUMENT; }
  std::vector<smhip_ndt_gicp_stats> st((size_t)K);
  for (auto& x : st) std::memset(&x, 0, sizeof(x));
  // smhip_set_target_cache(h, 0) = nothing survives from one Align to the next, as in the reference; INSIDE an Align the
  // target's structures are still built once (PCL builds its kd-trees in setInputTarget, not per iteration)
  struct WithinAlign {
    smhip_context* h; decltype(smhip_context::target_cache) keep;
    ~WithinAlign() { h->target_cache = keep; }
  } within{h, h->target_cache};
  if (!within.keep) {
    for (int e = 0; e < K; ++e) { g.job[first + e].staged_raw_gen = 0; g.job[first + e].cov_gen = 0; touch_grid(h, first + e, 1); }   // (cov_gen 0: a new epoch)
    for (auto& m : ndt_of(h).meta) m.valid = false;
    h->target_cache = 1;
  }
  // (see GicpCell) can any job have to estimate target covariances in this Align?
  bool all_full = true;
  for (int e = 0; e < K; ++e) {
    const GicpJobHost& jh = g.job[first + e];
    const int filt = g.opts.using_voxel_filter ? 1 : 0;
    const bool staged = jh.raw_tgt_gen != 0 && jh.staged_raw_gen == jh.raw_tgt_gen && jh.staged_slot_gen == h->tgt_gen[first + e] &&
                        jh.staged_filter == filt && (!filt || jh.staged_res == g.opts.voxel_resolution);
    const bool cov = jh.epoch != 0 && jh.cov_full && jh.cov_gen == h->tgt_gen[first + e] && jh.cov_k == g.opts.gicp_k_correspondences &&
                     jh.cov_eps == g.opts.gicp_epsilon;
    all_full = all_full && staged && cov;
  }
  GicpCell cell(h, all_full ? h->dev.grid_cell : gicp_knn_cell(g.opts));
  if (g.opts.using_voxel_filter) {
    s = ndt_gicp_stage_clouds_batch(h, first, K);
    if (s) return s;
  } else {
    for (int e = 0; e < K; ++e) {
      s = ndt_gicp_stage_clouds(h, first + e);
      if (s) return s;
    }
  }
  for (int e = 0; e < K; ++e) { st[e].n_source = h->ns[first + e]; st[e].n_target = h->nt[first + e]; }
  std::vector<float> ndt_guess((size_t)16 * K), fin((size_t)16 * K);
  std::vector<double> ndt_score((size_t)K, 0.9);             // :81
  for (int e = 0; e < K; ++e) colmajor_to_rm_f32(guesses + 16 * e, &ndt_guess[(size_t)16 * e]);   // guess.cast<float>(), :80
  if (g.opts.use_ndt) {                                      // :82-90
    NdtHost& n = ndt_of(h);
    const smhip_ndt_options saved = n.opts;
    const bool dm = n.double_math;
    smhip_ndt_options no;
    smhip_ndt_default_options(&no);
    no.resolution = g.opts.ndt_resolution; no.step_size = g.opts.ndt_step_size;
    no.transformation_epsilon = g.opts.ndt_transformation_epsilon; no.max_iterations = g.opts.ndt_max_iterations;
    n.opts = no; n.double_math = true;
    std::vector<double> nres((size_t)16 * K);
    std::vector<smhip_ndt_stats> ns((size_t)K);
    for (auto& x : ns) std::memset(&x, 0, sizeof(x));
    s = ndt_align_slots(h, first, K, guesses, nres.data(), ndt_score.data(), ns.data());
    n.opts = saved; n.double_math = dm;
    if (s) return s;
    for (int e = 0; e < K; ++e) {
      colmajor_to_rm_f32(&nres[(size_t)16 * e], &ndt_guess[(size_t)16 * e]);     // ndt_.getFinalTransformation()
      st[e].ndt_iterations = ns[e].iterations;
    }
  }
  std::vector<char> run((size_t)K, 0);
  bool any = false;
  for (int e = 0; e < K; ++e) { st[e].ndt_score = ndt_score[e]; run[e] = ndt_score[e] <= 1.0 ? 1 : 0; any = any || run[e]; }   // :94
  std::vector<double> fit((size_t)K, 10.0);                  // :92
  if (any) {
    s = gicp_align_jobs(h, first, K, run.data(), ndt_guess.data(), fin.data(), st.data());
    if (s) return s;
  }
  for (int e = 0; e < K; ++e) {
    if (run[e]) rm_f32_to_colmajor(&fin[(size_t)16 * e], results + 16 * e);
    else for (int i = 0; i < 16; ++i) results[16 * e + i] = guesses[16 * e + i];   // :106
  }
  if (any) {
    std::vector<double> f2((size_t)K);
    s = fitness_scores(h, first, K, results, f2.data());    // gicp_.getFitnessScore(), :101
    if (s) return s;
    for (int e = 0; e < K; ++e) if (run[e]) fit[e] = f2[e];
  }
  for (int e = 0; e < K; ++e) {
    st[e].ok = run[e] ? 1 : 0;
    st[e].gicp_score = fit[e];
    if (scores) scores[e] = std::exp(-fit[e]);               // :103 / :107
    if (stats) stats[e] = st[e];
  }
  return SMHIP_OK;
}

smhip_status smhip_ndt_gicp_align(smhip_handle h, const double guess[16], double result[16], double* score, smhip_ndt_gicp_stats* stats) {
  if (!h || !guess || !result) return SMHIP_ERR_INVALID_ARGUMENT;
  return ndt_gicp_align_jobs(h, 0, 1, guess, result, score, stats);
}

smhip_status smhip_ndt_gicp_align_batch(smhip_handle h, int first_job, int njobs, const double* guesses, double* results, double* scores,
                                        smhip_ndt_gicp_stats* stats) {
  if (!h || !guesses || !results) return SMHIP_ERR_INVALID_ARGUMENT;
  return ndt_gicp_align_jobs(h, first_job, njobs, guesses, results, scores, stats);
}

// GICP alone on slot 0's clouds as set by smhip_set_source_f32 / smhip_set_target_f32 (parity hook; also what a
// caller that wants plain pcl GICP semantics uses).  *fitness = getFitnessScore().
smhip_status smhip_gicp_align(smhip_handle h, const double guess[16], double result[16], double* fitness, smhip_ndt_gicp_stats* stats) {
  if (!h || !guess || !result) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = gicp_ensure(h);
  if (s) return s;
  if (h->ns[0] <= 0 || h->nt[0] <= 0) { h->err = "GICP before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
  smhip_ndt_gicp_stats st{};
  st.n_source = h->ns[0]; st.n_target = h->nt[0];
  float g32[16], fin[16];
  colmajor_to_rm_f32(guess, g32);
  GicpCell cell(h, gicp_knn_cell(gicp_of(h).opts));
  const char run = 1;
  s = gicp_align_jobs(h, 0, 1, &run, g32, fin, &st);
  if (s) return s;
  rm_f32_to_colmajor(fin, result);
  double fit = 0;
  s = fitness_score(h, result, &fit);
  if (s) return s;
  st.ok = 1; st.gicp_score = fit;
  if (fitness) *fitness = fit;
  if (stats) *stats = st;
  return SMHIP_OK;
}

// parity hook: the GICP functor (f and its 6-gradient, gicp_omp_impl.hpp:250-377) at state x over the
// correspondences of the LAST outer iteration of the last GICP run, with base_transformation_ = guess
smhip_status smhip_gicp_evaluate(smhip_handle h, const double guess[16], const double x[6], double* f, double grad[6]) {
  if (!h || !guess || !x || !f || !grad) return SMHIP_ERR_INVALID_ARGUMENT;
  GicpHost& g = gicp_of(h);
  if (!g.allocated || h->ns[0] <= 0) { h->err = "GICP has not run"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  // one functor evaluation of job 0, outside any run: the "task" is this call
  GicpTask t;
  t.h = h; t.job = 0; t.ns = h->ns[0];
  colmajor_to_rm_f32(guess, t.guess);
  float T[16];
  apply_state_f32(t.guess, x, T);
  for (int i = 0; i < 12; ++i) { t.P.T[i] = T[i]; t.P.B[i] = t.guess[i]; }
  std::vector<GicpTask*> none, one{&t};
  const smhip_status s = gicp_round(h, 0, 1, none, one);
  if (s) return s;
  const double* o = g.out_pinned;
  const double m = o[13];
  *f = o[0] / m;
  for (int i = 0; i < 3; ++i) grad[i] = o[1 + i] * (2.0 / m);
  double R[9];
  for (int i = 0; i < 9; ++i) R[i] = o[4 + i] * (2.0 / m);
  r_derivative(x, R, grad);
  return SMHIP_OK;
}

// parity hooks -----------------------------------------------------------------------------------
// the working clouds of the last smhip_ndt_gicp_align (which = 0 source, 1 target) in filter output order
smhip_status smhip_ndt_gicp_get_downsampled(smhip_handle h, int which, float* xyz, int capacity, int* n_out) {
  if (!h || !n_out || (which != 0 && which != 1)) return SMHIP_ERR_INVALID_ARGUMENT;
  const int n = which == 0 ? h->ns[0] : h->nt[0];
  *n_out = n;
  if (!xyz) return SMHIP_OK;
  if (capacity < n) { h->err = "capacity too small"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<float4> tmp((size_t)n);
  HIPCHK(h, hipMemcpyAsync(tmp.data(), which == 0 ? h->dev.src : h->dev.tgt_p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    // the source was Morton-ordered after the filter: w holds its position in the filter's output
    const int k = which == 0 ? [&] { int v; std::memcpy(&v, &tmp[i].w, 4); return v; }() : i;
    xyz[3 * k] = tmp[i].x; xyz[3 * k + 1] = tmp[i].y; xyz[3 * k + 2] = tmp[i].z;
  }
  return SMHIP_OK;
}

// covariances of the last GICP run in the order of smhip_ndt_gicp_get_downsampled / the uploaded clouds:
// cov = n x 6 doubles (xx xy xz yy yz zz)
smhip_status smhip_gicp_get_covariances(smhip_handle h, int which, double* cov, int n) {
  if (!h || !cov || (which != 0 && which != 1)) return SMHIP_ERR_INVALID_ARGUMENT;
  GicpHost& g = gicp_of(h);
  if (!g.allocated) { h->err = "GICP has not run"; return SMHIP_ERR_NOT_READY; }
  const int m = which == 0 ? h->ns[0] : h->nt[0];
  if (n != m) { h->err = "n must equal the cloud size"; return SMHIP_ERR_INVALID_ARGUMENT; }
  HIPCHK(h, hipSetDevice(h->device));
  if (which == 1) {
    // a run estimates a target's covariances only where a source point was matched: the hook asks for all of them, as
    // computeCovariances would have left them (:391-402) -- the same function over the job slot's grid, every point
    const smhip_ndt_gicp_options& o = g.opts;
    GicpSearchMode mode(h);
    GicpCell cell(h, gicp_knn_cell(o));
    bool kept = false;
    smhip_status s = gicp_build_grids(h, 0, 1, &kept);
    if (s) return s;
    IcpDev dk = h->dev;
    dk.have_rowbits = 1;
    GicpKnnBatch L{};
    L.n = 1; L.slot[0] = 0; L.cov[0] = g.dev.cov_t;
    const int k = o.gicp_k_correspondences;
    if (k <= 20) hipLaunchKernelGGL(gicp_knn_cov<20>, dim3(ceil_div(m, kGicpKnnThreads), 1), dim3(kGicpKnnThreads), 0, h->stream, dk, L, k, o.gicp_epsilon);
    else hipLaunchKernelGGL(gicp_knn_cov<kGicpKMax>, dim3(ceil_div(m, kGicpKnnThreads), 1), dim3(kGicpKnnThreads), 0, h->stream, dk, L, k, o.gicp_epsilon);
    HIPCHK(h, hipGetLastError());
  }
  std::vector<double> c((size_t)m * 6);
  HIPCHK(h, hipMemcpyAsync(c.data(), which == 0 ? g.dev.cov_s : g.dev.cov_t, sizeof(double) * 6 * (size_t)m, hipMemcpyDeviceToHost, h->stream));
  if (which == 1) {                                          // already indexed by the target's array position
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(cov, c.data(), sizeof(double) * 6 * (size_t)m);
    return SMHIP_OK;
  }
  std::vector<float4> pts((size_t)m);
  HIPCHK(h, hipMemcpyAsync(pts.data(), h->dev.src, sizeof(float4) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < m; ++i) {
    int k; std::memcpy(&k, &pts[i].w, 4);                    // caller / filter-output index of source entry i
    for (int e = 0; e < 6; ++e) cov[(size_t)6 * k + e] = c[(size_t)6 * i + e];
  }
  return SMHIP_OK;
}

}  // extern "C"
