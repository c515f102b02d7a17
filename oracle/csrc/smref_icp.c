/* smref_icp.c -- dependency-free C restatement of registrators::IcpFast and
 * EigenPointCloud::CalculateNormals.  TEST ORACLE / TIMED CPU BASELINE ONLY;
 * never linked into the product (see oracle/__init__.py).  PARITY UNPINNED:
 * the reference has no registrator tests and cannot be built here.
 *
 * Restates (paths relative to /root/reference):
 *   registrators/icp_fast.cc:65-90    Matches::GetDistsQuantile
 *   registrators/icp_fast.cc:100-166  ErrorElements (compaction + gather)
 *   registrators/icp_fast.cc:169-180  FindClosests: exact 1-NN (own kd-tree, smallest-id tie rule) or, with
 *                                     nn_eps >= 0, libnabo's own search: the KDTREE_LINEAR_HEAP tree of
 *                                     libnabo tags/1.0.7 (setup/install_libnabo.sh:16-18; not vendored by the
 *                                     reference) restated from its published source nabo/kdtree_cpu.cpp --
 *                                     buildNodes (bucketSize 8, argMax of the INHERITED box, nth_element at
 *                                     count - count/2, cutVal = that element) and recurseKnn (incremental
 *                                     off[]/rd bound, far side pruned unless rd * (1+eps)^2 < best).  With
 *                                     nn_eps = 3.16 this is the reference's call at icp_fast.cc:174-178.
 *   registrators/icp_fast.cc:204-324  point-to-plane normal equations + solve
 *   registrators/icp_fast.cc:377-405  CheckConvergence
 *   registrators/icp_fast.cc:455-529  IcpFast::Align
 *   builder/data/cloud_types.cc:73-144, 347-368  CalculateNormals
 * Matrices crossing this C API are ROW-major 4x4 doubles (numpy default).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "smref_internal.h"

double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* exact kd-tree (replaces libnabo KDTREE_LINEAR_HEAP, icp_fast.cc:466) */
/* ------------------------------------------------------------------ */
#define KD_LEAF 8

void kd_select(const double* pts, int* idx, int lo, int hi, int k, int dim) {
  /* quickselect so that idx[k] holds the k-th smallest by coordinate dim */
  while (hi - lo > 1) {
    double pivot = pts[3 * idx[lo + (hi - lo) / 2] + dim];
    int i = lo, j = hi - 1;
    while (i <= j) {
      while (pts[3 * idx[i] + dim] < pivot) ++i;
      while (pts[3 * idx[j] + dim] > pivot) --j;
      if (i <= j) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; ++i; --j; }
    }
    if (k <= j) hi = j + 1;
    else if (k >= i) lo = i;
    else return;
  }
}

static int kd_build_rec(KdTree* t, int lo, int hi) {
  int id = t->n_nodes++;
  t->node_lo[id] = lo; t->node_hi[id] = hi;
  if (hi - lo <= KD_LEAF) { t->node_dim[id] = -1; return id; }
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = lo; i < hi; ++i)
    for (int d = 0; d < 3; ++d) {
      double v = t->pts[3 * t->perm[i] + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  int dim = 0;
  if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
  if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
  int mid = (lo + hi) / 2;
  kd_select(t->pts, t->perm, lo, hi, mid, dim);
  t->node_dim[id] = dim;
  t->node_cut[id] = t->pts[3 * t->perm[mid] + dim];
  int l = kd_build_rec(t, lo, mid);
  int r = kd_build_rec(t, mid, hi);
  t->node_left[id] = l; t->node_right[id] = r;
  return id;
}

KdTree* kd_build(const double* pts, int n) {
  KdTree* t = (KdTree*)calloc(1, sizeof(KdTree));
  t->n = n; t->pts = pts;
  t->cap_nodes = 2 * (n / (KD_LEAF / 2) + 2) + 8;
  t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  t->node_lo = (int*)malloc(sizeof(int) * t->cap_nodes);
  t->node_hi = (int*)malloc(sizeof(int) * t->cap_nodes);
  t->node_dim = (int*)malloc(sizeof(int) * t->cap_nodes);
  t->node_left = (int*)malloc(sizeof(int) * t->cap_nodes);
  t->node_right = (int*)malloc(sizeof(int) * t->cap_nodes);
  t->node_cut = (double*)malloc(sizeof(double) * t->cap_nodes);
  for (int i = 0; i < n; ++i) t->perm[i] = i;
  if (n > 0) kd_build_rec(t, 0, n);
  return t;
}

void kd_free(KdTree* t) {
  free(t->perm); free(t->node_lo); free(t->node_hi); free(t->node_dim);
  free(t->node_left); free(t->node_right); free(t->node_cut); free(t);
}

void kd_nn(const KdTree* t, const double q[3], int* best_id, double* best_d2) {
  int stack_node[64]; double stack_d[64]; int sp = 0;
  double bd = INFINITY; int bi = -1;
  if (t->n == 0) { *best_id = -1; *best_d2 = INFINITY; return; }
  stack_node[sp] = 0; stack_d[sp++] = 0.0;
  while (sp > 0) {
    int id = stack_node[--sp];
    if (stack_d[sp] >= bd) continue;
    while (t->node_dim[id] >= 0) {
      int dim = t->node_dim[id];
      double diff = q[dim] - t->node_cut[id];
      int near = diff < 0 ? t->node_left[id] : t->node_right[id];
      int far = diff < 0 ? t->node_right[id] : t->node_left[id];
      if (diff * diff < bd && sp < 64) { stack_node[sp] = far; stack_d[sp++] = diff * diff; }
      id = near;
    }
    for (int i = t->node_lo[id]; i < t->node_hi[id]; ++i) {
      int p = t->perm[i];
      double dx = q[0] - t->pts[3 * p], dy = q[1] - t->pts[3 * p + 1], dz = q[2] - t->pts[3 * p + 2];
      double d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < bd || (d2 == bd && p < bi)) { bd = d2; bi = p; }
    }
  }
  *best_id = bi; *best_d2 = bd;
}


/* ------------------------------------------------------------------ */
/* libnabo 1.0.7 KDTreeUnbalancedPtInLeavesImplicitBoundsStackOpt       */
/* (what NNS::create(..., NNS::KDTREE_LINEAR_HEAP) builds, icp_fast.cc:466, and what knn(..., epsilon, ...) */
/*  walks, icp_fast.cc:177-178).  Restated from the library's published source; the library is absent here. */
/* ------------------------------------------------------------------ */
#define NABO_BUCKET 8            /* additionalParameters "bucketSize" default */
typedef struct {
  const double* pts;             /* [n][3] */
  int* perm;                     /* build points, permuted by nth_element */
  int* dim;                      /* per node: cut dimension, 3 = leaf */
  double* cut;                   /* per node: cutVal */
  int* child;                    /* internal: right child (left child = node + 1); leaf: first bucket entry */
  int* count;                    /* leaf: bucket size */
  int n_nodes, cap;
} NaboTree;

static int nabo_build_nodes(NaboTree* t, int first, int last, const double mn[3], const double mx[3]) {
  const int cnt = last - first;
  const int pos = t->n_nodes++;
  if (cnt <= NABO_BUCKET) {                                   /* bucket: entries in build order */
    t->dim[pos] = 3; t->child[pos] = first; t->count[pos] = cnt; t->cut[pos] = 0;
    return pos;
  }
  int cd = 0; double mv = 0.0;                                /* argMax: starts from (0, 0.) */
  for (int i = 0; i < 3; ++i) if (mx[i] - mn[i] > mv) { mv = mx[i] - mn[i]; cd = i; }
  const int right = cnt / 2, left = cnt - right;
  kd_select(t->pts, t->perm, first, last, first + left, cd);  /* std::nth_element(first, first + leftCount, last) */
  const double cv = t->pts[3 * t->perm[first + left] + cd];
  double lmx[3] = {mx[0], mx[1], mx[2]}, rmn[3] = {mn[0], mn[1], mn[2]};
  lmx[cd] = cv; rmn[cd] = cv;
  t->dim[pos] = cd; t->cut[pos] = cv;
  nabo_build_nodes(t, first, first + left, mn, lmx);          /* left child == pos + 1 */
  t->child[pos] = nabo_build_nodes(t, first + left, last, rmn, mx);
  return pos;
}

static NaboTree* nabo_build(const double* pts, int n) {
  NaboTree* t = (NaboTree*)calloc(1, sizeof(NaboTree));
  t->pts = pts;
  t->cap = 2 * (n / (NABO_BUCKET / 2) + 2) + 8;
  t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  t->dim = (int*)malloc(sizeof(int) * t->cap); t->child = (int*)malloc(sizeof(int) * t->cap);
  t->count = (int*)malloc(sizeof(int) * t->cap); t->cut = (double*)malloc(sizeof(double) * t->cap);
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i) {
    t->perm[i] = i;
    for (int d = 0; d < 3; ++d) { if (pts[3 * i + d] < mn[d]) mn[d] = pts[3 * i + d]; if (pts[3 * i + d] > mx[d]) mx[d] = pts[3 * i + d]; }
  }
  if (n > 0) nabo_build_nodes(t, 0, n, mn, mx);
  return t;
}
static void nabo_free(NaboTree* t) { free(t->perm); free(t->dim); free(t->child); free(t->count); free(t->cut); free(t); }

/* recurseKnn for k = 1, maxRadius = inf, ALLOW_SELF_MATCH: head of the heap = (best_id, best_d2) */
static void nabo_recurse(const NaboTree* t, const double* q, int n, double rd, double off[3], double max_error2,
                         int* best_id, double* best_d2, long* leaves) {
  const int cd = t->dim[n];
  if (cd == 3) {
    const int* e = t->perm + t->child[n];
    for (int i = 0; i < t->count[n]; ++i) {
      const double* p = t->pts + 3 * e[i];
      double dist = 0;
      for (int d = 0; d < 3; ++d) { const double diff = q[d] - p[d]; dist += diff * diff; }
      if (dist < *best_d2) { *best_d2 = dist; *best_id = e[i]; }
    }
    if (leaves) ++*leaves;
    return;
  }
  const int right = t->child[n];
  const double old_off = off[cd], new_off = q[cd] - t->cut[n];
  if (new_off > 0) {
    nabo_recurse(t, q, right, rd, off, max_error2, best_id, best_d2, leaves);
    rd += -old_off * old_off + new_off * new_off;
    if (rd * max_error2 < *best_d2) {
      off[cd] = new_off;
      nabo_recurse(t, q, n + 1, rd, off, max_error2, best_id, best_d2, leaves);
      off[cd] = old_off;
    }
  } else {
    nabo_recurse(t, q, n + 1, rd, off, max_error2, best_id, best_d2, leaves);
    rd += -old_off * old_off + new_off * new_off;
    if (rd * max_error2 < *best_d2) {
      off[cd] = new_off;
      nabo_recurse(t, q, right, rd, off, max_error2, best_id, best_d2, leaves);
      off[cd] = old_off;
    }
  }
}
static void nabo_nn(const NaboTree* t, const double q[3], double eps, int* best_id, double* best_d2, long* leaves) {
  double off[3] = {0, 0, 0};
  *best_id = -1; *best_d2 = INFINITY;
  if (t->n_nodes == 0) return;
  nabo_recurse(t, q, 0, 0.0, off, (1 + eps) * (1 + eps), best_id, best_d2, leaves);
}

/* ------------------------------------------------------------------ */
/* small dense helpers                                                  */
/* ------------------------------------------------------------------ */
static void mat4_mul(const double* a, const double* b, double* c) { /* row-major */
  double r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      r[4 * i + j] = s;
    }
  memcpy(c, r, sizeof(r));
}
static void mat4_eye(double* a) { memset(a, 0, 16 * sizeof(double)); a[0] = a[5] = a[10] = a[15] = 1.0; }

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 6) */
void jacobi_eig(int n, double* A /*n*n, destroyed*/, double* V, double* w) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[n * i + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[n * i + j] * A[n * i + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[n * p + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[n * q + q] - A[n * p + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = A[n * k + p], akq = A[n * k + q];
          A[n * k + p] = c * akp - s * akq; A[n * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = A[n * p + k], aqk = A[n * q + k];
          A[n * p + k] = c * apk - s * aqk; A[n * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[n * k + p], vkq = V[n * k + q];
          V[n * k + p] = c * vkp - s * vkq; V[n * k + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[n * i + i];
}

/* icp_fast.cc:204-254: invertible -> Cholesky; else min-norm (pseudo-inverse) */
static void solve_possibly_underdetermined6(const double* A, const double* b, double* x) {
  double E[36], V[36], w[6];
  memcpy(E, A, sizeof(E));
  jacobi_eig(6, E, V, w);
  double wmax = 0;
  for (int i = 0; i < 6; ++i) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
  double thresh = 2.220446049250313e-16 * 6 * wmax;
  int rank = 0;
  for (int i = 0; i < 6; ++i) if (fabs(w[i]) > thresh) ++rank;
  if (rank == 6) {
    double L[36];
    int ok = 1;
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6 && ok; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = A[6 * i + j];
        for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
        if (i == j) { if (s <= 0) { ok = 0; break; } L[6 * i + i] = sqrt(s); }
        else L[6 * i + j] = s / L[6 * j + j];
      }
    if (ok) {
      double y[6];
      for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
      for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
      return;
    }
  }
  for (int i = 0; i < 6; ++i) x[i] = 0;
  for (int k = 0; k < 6; ++k) {
    if (fabs(w[k]) <= thresh) continue;
    double c = 0;
    for (int i = 0; i < 6; ++i) c += V[6 * i + k] * b[i];
    c /= w[k];
    for (int i = 0; i < 6; ++i) x[i] += c * V[6 * i + k];
  }
}

static void angle_axis_to_T(const double x[6], double T[16]) { /* icp_fast.cc:306-321 */
  mat4_eye(T);
  double ang = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double ax = x[0] / ang, ay = x[1] / ang, az = x[2] / ang;
  double c = cos(ang), s = sin(ang), v = 1 - c;
  double R[9] = {c + v * ax * ax, v * ax * ay - s * az, v * ax * az + s * ay,
                 v * ax * ay + s * az, c + v * ay * ay, v * ay * az - s * ax,
                 v * ax * az - s * ay, v * ay * az + s * ax, c + v * az * az};
  int nan = 0;
  for (int i = 0; i < 9; ++i) if (isnan(R[i])) nan = 1;
  for (int i = 0; i < 3; ++i) if (isnan(x[3 + i])) nan = 1;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[4 * i + j] = nan ? (i == j) : R[3 * i + j];
  T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
}

static void quat_from_R(const double* T /*row-major 4x4*/, double q[4]) {
  double r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[4], r11 = T[5], r12 = T[6], r20 = T[8], r21 = T[9], r22 = T[10];
  double tr = r00 + r11 + r22;
  if (tr > 0) {
    double s = sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (r21 - r12) / s; q[2] = (r02 - r20) / s; q[3] = (r10 - r01) / s;
  } else if (r00 >= r11 && r00 >= r22) {
    double s = sqrt(1.0 + r00 - r11 - r22) * 2;
    q[0] = (r21 - r12) / s; q[1] = 0.25 * s; q[2] = (r01 + r10) / s; q[3] = (r02 + r20) / s;
  } else if (r11 >= r22) {
    double s = sqrt(1.0 + r11 - r00 - r22) * 2;
    q[0] = (r02 - r20) / s; q[1] = (r01 + r10) / s; q[2] = 0.25 * s; q[3] = (r12 + r21) / s;
  } else {
    double s = sqrt(1.0 + r22 - r00 - r11) * 2;
    q[0] = (r10 - r01) / s; q[1] = (r02 + r20) / s; q[2] = (r12 + r21) / s; q[3] = 0.25 * s;
  }
}
static double quat_angdist(const double a[4], const double b[4]) {
  /* a * conj(b) */
  double w = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  double x = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
  double y = -a[0] * b[2] + a[2] * b[0] - a[3] * b[1] + a[1] * b[3];
  double z = -a[0] * b[3] + a[3] * b[0] - a[1] * b[2] + a[2] * b[1];
  return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

/* nth_element rank rule on a scratch copy (icp_fast.cc:65-90) */
static double select_kth(double* v, int n, int k) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    double pivot = v[lo + (hi - lo) / 2];
    int i = lo, j = hi - 1;
    while (i <= j) {
      while (v[i] < pivot) ++i;
      while (v[j] > pivot) --j;
      if (i <= j) { double t = v[i]; v[i] = v[j]; v[j] = t; ++i; --j; }
    }
    if (k <= j) hi = j + 1;
    else if (k >= i) lo = i;
    else break;
  }
  return v[k];
}

/* ------------------------------------------------------------------ */
/* IcpFast::Align                                                       */
/* ------------------------------------------------------------------ */
/* block_times[6]: FindClosests, GetDistsQuantile+ErrorElements, ComputePointToPlane, BuildKdTree, ApplyTransform,
 * whole Align (names follow the reference's REGISTER_BLOCK labels, icp_fast.cc:103,171,261,464,484).
 * nn_eps < 0: exact 1-NN with the smallest-id tie rule (own kd-tree; what the GPU path implements);
 * nn_eps >= 0: libnabo's tree and eps-approximate search (3.16 = the reference's call, icp_fast.cc:174).
 * nthreads: FindClosests runs under OpenMP like libnabo's knn (schedule(guided, 32)); ApplyTransform and the normal
 * equations are parallel too (the reference is built with -fopenmp, CMakeLists.txt:17-28, so Eigen's products are);
 * the sums are folded in thread order, so results depend on nthreads only at the 1e-16 level. */
int smref_icp_align_ex(const double* src, int ns, const double* tgt, const double* nrm, int nt,
                       const double* guess, int max_iteration, float dist_outlier_ratio,
                       int early_exit, int nthreads, double nn_eps, double* result, double* score, int* iterations,
                       double* block_times, int* last_ids, double* last_d2, long* leaves_visited) {
  if (ns <= 0 || nt <= 0) return -1;
  if (nthreads < 1) nthreads = 1;
  const double t_begin = now_s();
  double t_nn = 0, t_err = 0, t_p2p = 0, t_kd = 0, t_tf = 0, t0;
  double mu[3] = {0, 0, 0};
  for (int j = 0; j < nt; ++j) for (int d = 0; d < 3; ++d) mu[d] += tgt[3 * j + d];   /* :457-458 */
  for (int d = 0; d < 3; ++d) mu[d] /= nt;
  double* Q = (double*)malloc(sizeof(double) * 3 * (size_t)nt);
  for (int j = 0; j < nt; ++j) for (int d = 0; d < 3; ++d) Q[3 * j + d] = tgt[3 * j + d] - mu[d];  /* :462 */
  t0 = now_s();
  KdTree* tree = NULL; NaboTree* nabo = NULL;
  if (nn_eps < 0) tree = kd_build(Q, nt); else nabo = nabo_build(Q, nt);                /* :464-467 */
  t_kd = now_s() - t0;
  double Tm[16], Tmi[16], G[16], Titer[16];
  mat4_eye(Tm); Tm[3] = mu[0]; Tm[7] = mu[1]; Tm[11] = mu[2];
  mat4_eye(Tmi); Tmi[3] = -mu[0]; Tmi[7] = -mu[1]; Tmi[11] = -mu[2];
  mat4_mul(Tmi, guess, G);                                                            /* :469 */
  double* P0 = (double*)malloc(sizeof(double) * 3 * (size_t)ns);
  double* P = (double*)malloc(sizeof(double) * 3 * (size_t)ns);
  int* ids = (int*)malloc(sizeof(int) * (size_t)ns);
  double* d2 = (double*)malloc(sizeof(double) * (size_t)ns);
  double* scratch = (double*)malloc(sizeof(double) * (size_t)ns);
  double* part = (double*)calloc((size_t)nthreads * 48, sizeof(double));                /* per-thread A(36) b(6) sum kept */
  for (int i = 0; i < ns; ++i)                                                        /* :470 */
    for (int r = 0; r < 3; ++r)
      P0[3 * i + r] = G[4 * r] * src[3 * i] + G[4 * r + 1] * src[3 * i + 1] + G[4 * r + 2] * src[3 * i + 2] + G[4 * r + 3];
  mat4_eye(Titer);                                                                    /* :473 */
  double (*rots)[4] = (double(*)[4])malloc(sizeof(double) * 4 * (size_t)(max_iteration + 2));
  double (*trs)[3] = (double(*)[3])malloc(sizeof(double) * 3 * (size_t)(max_iteration + 2));
  rots[0][0] = 1; rots[0][1] = rots[0][2] = rots[0][3] = 0;                           /* :478-479 */
  trs[0][0] = trs[0][1] = trs[0][2] = 0;
  int nh = 1, it = 0;
  long leaves = 0;
  const double rho = (double)dist_outlier_ratio;                                      /* float option widened */
  for (;;) {
    t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int i = 0; i < ns; ++i)                                                      /* :486-491 */
      for (int r = 0; r < 3; ++r)
        P[3 * i + r] = Titer[4 * r] * P0[3 * i] + Titer[4 * r + 1] * P0[3 * i + 1] + Titer[4 * r + 2] * P0[3 * i + 2] + Titer[4 * r + 3];
    t_tf += now_s() - t0;
    t0 = now_s();
    if (tree) {
#ifdef _OPENMP
#pragma omp parallel for schedule(guided, 32) num_threads(nthreads)
#endif
      for (int i = 0; i < ns; ++i) kd_nn(tree, &P[3 * i], &ids[i], &d2[i]);           /* :493 */
    } else {
      long lv = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(guided, 32) num_threads(nthreads) reduction(+ : lv)
#endif
      for (int i = 0; i < ns; ++i) nabo_nn(nabo, &P[3 * i], nn_eps, &ids[i], &d2[i], &lv);
      leaves += lv;
    }
    t_nn += now_s() - t0;
    t0 = now_s();
    int nv = 0;
    for (int i = 0; i < ns; ++i) if (d2[i] != INFINITY) scratch[nv++] = d2[i];        /* :71-77 */
    if (nv == 0) {
      free(Q); free(P0); free(P); free(ids); free(d2); free(scratch); free(part); free(rots); free(trs);
      if (tree) kd_free(tree);
      if (nabo) nabo_free(nabo);
      return -2;
    }
    double limit;
    if (rho == 1.0) { limit = scratch[0]; for (int i = 1; i < nv; ++i) if (scratch[i] > limit) limit = scratch[i]; }
    else { int k = (int)(nv * rho); limit = select_kth(scratch, nv, k); }               /* :86-89 */
    t_err += now_s() - t0;
    t0 = now_s();
    memset(part, 0, sizeof(double) * (size_t)nthreads * 48);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
      const int tid = 0, nth = 1;
#endif
      double* A = part + 48 * (size_t)tid;
      double* b = A + 36;
      const int lo = (int)((long)ns * tid / nth), hi = (int)((long)ns * (tid + 1) / nth);
      for (int i = lo; i < hi; ++i) {
        if (!(d2[i] <= limit) || d2[i] == INFINITY) continue;                          /* :497-498, :124-128 */
        const double* p = &P[3 * i]; const double* q = &Q[3 * ids[i]]; const double* n = &nrm[3 * ids[i]];
        double J[6] = {p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2], p[0] * n[1] - p[1] * n[0], n[0], n[1], n[2]};
        double r = (p[0] - q[0]) * n[0] + (p[1] - q[1]) * n[1] + (p[2] - q[2]) * n[2];  /* :293-299 */
        for (int a = 0; a < 6; ++a) { for (int c = 0; c < 6; ++c) A[6 * a + c] += J[a] * J[c]; b[a] -= J[a] * r; }
        A[42] += sqrt(d2[i]); A[43] += 1.0;
      }
    }
    double A[36], b[6], sum_sqrt = 0; int kept = 0;
    memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b));
    for (int t = 0; t < nthreads; ++t) {
      const double* pa = part + 48 * (size_t)t;
      for (int k = 0; k < 36; ++k) A[k] += pa[k];
      for (int k = 0; k < 6; ++k) b[k] += pa[36 + k];
      sum_sqrt += pa[42]; kept += (int)pa[43];
    }
    double x[6], dT[16];
    solve_possibly_underdetermined6(A, b, x);                                          /* :304 */
    angle_axis_to_T(x, dT);                                                            /* :306-321 */
    mat4_mul(dT, Titer, Titer);                                                        /* :506-510 */
    t_p2p += now_s() - t0;
    ++it;                                                                              /* :513 */
    quat_from_R(Titer, rots[nh]); trs[nh][0] = Titer[3]; trs[nh][1] = Titer[7]; trs[nh][2] = Titer[11]; ++nh;
    int conv = 0;
    if (early_exit && nh > 4) {                                                        /* :377-405 */
      double rd = 0, td = 0;
      for (int i = nh - 1; i >= nh - 4; --i) {
        rd += fabs(quat_angdist(rots[i], rots[i - 1]));
        double dx = trs[i][0] - trs[i - 1][0], dy = trs[i][1] - trs[i - 1][1], dz = trs[i][2] - trs[i - 1][2];
        td += sqrt(dx * dx + dy * dy + dz * dz);
      }
      conv = (rd / 4 < 1e-3) && (td / 4 < 1e-2);
    }
    if (conv || it >= max_iteration) {                                                 /* :516-522 */
      *score = exp(-sum_sqrt / kept);
      break;
    }
  }
  double tmp[16];
  mat4_mul(Titer, G, tmp); mat4_mul(Tm, tmp, result);                                  /* :527 */
  *iterations = it;
  if (block_times) { block_times[0] = t_nn; block_times[1] = t_err; block_times[2] = t_p2p; block_times[3] = t_kd;
                     block_times[4] = t_tf; block_times[5] = now_s() - t_begin; }
  if (last_ids) memcpy(last_ids, ids, sizeof(int) * (size_t)ns);
  if (last_d2) memcpy(last_d2, d2, sizeof(double) * (size_t)ns);
  if (leaves_visited) *leaves_visited = leaves;
  free(Q); free(P0); free(P); free(ids); free(d2); free(scratch); free(part); free(rots); free(trs);
  if (tree) kd_free(tree);
  if (nabo) nabo_free(nabo);
  return 0;
}

/* eps-approximate 1-NN through the libnabo restatement (eps >= 0), for tests of the search alone */
int smref_nn_nabo(const double* tgt, int nt, const double* qry, int nq, double eps, int* ids, double* d2, long* leaves) {
  NaboTree* t = nabo_build(tgt, nt);
  long lv = 0;
  for (int i = 0; i < nq; ++i) nabo_nn(t, &qry[3 * i], eps, &ids[i], &d2[i], &lv);
  if (leaves) *leaves = lv;
  nabo_free(t);
  return 0;
}

/* exact 1-NN of every query against a point set (for kernel-level parity tests) */
int smref_nn(const double* tgt, int nt, const double* qry, int nq, int* ids, double* d2) {
  KdTree* tree = kd_build(tgt, nt);
  for (int i = 0; i < nq; ++i) kd_nn(tree, &qry[3 * i], &ids[i], &d2[i]);
  kd_free(tree);
  return 0;
}

/* ------------------------------------------------------------------ */
/* EigenPointCloud::CalculateNormals  (cloud_types.cc:73-144, 347-368) */
/* ------------------------------------------------------------------ */
typedef struct { const double* pts; int* indices; double* out_p; double* out_n; int* out_k; int* out_sz; int m; } NormCtx;

static int rank3_sym(const double C[9]) {
  double E[9], V[9], w[3];
  memcpy(E, C, sizeof(E));
  jacobi_eig(3, E, V, w);
  double wmax = 0; int r = 0;
  for (int i = 0; i < 3; ++i) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
  for (int i = 0; i < 3; ++i) if (fabs(w[i]) > 2.220446049250313e-16 * 3 * wmax) ++r;
  return r;
}

static void normals_leaf(NormCtx* c, int first, int last) {                            /* :73-103 */
  int n = last - first;
  if (n <= 0) return;
  double M[9] = {0}, b[3] = {0};
  int kmin = c->indices[first];
  for (int i = first; i < last; ++i) {
    const double* p = &c->pts[3 * c->indices[i]];
    if (c->indices[i] < kmin) kmin = c->indices[i];
    for (int a = 0; a < 3; ++a) { b[a] += p[a]; for (int d = 0; d < 3; ++d) M[3 * a + d] += p[a] * p[d]; }
  }
  double mean[3] = {b[0] / n, b[1] / n, b[2] / n}, C[9] = {0};
  for (int i = first; i < last; ++i) {
    const double* p = &c->pts[3 * c->indices[i]];
    double e[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    for (int a = 0; a < 3; ++a) for (int d = 0; d < 3; ++d) C[3 * a + d] += e[a] * e[d];
  }
  if (rank3_sym(C) + 1 < 3) return;                                                     /* :90-92 */
  /* normal = M^-1 b via cofactors (Eigen 3x3 inverse), :94 */
  double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  double inv[9] = {c00 / det, (M[2] * M[7] - M[1] * M[8]) / det, (M[1] * M[5] - M[2] * M[4]) / det,
                   c01 / det, (M[0] * M[8] - M[2] * M[6]) / det, (M[2] * M[3] - M[0] * M[5]) / det,
                   c02 / det, (M[1] * M[6] - M[0] * M[7]) / det, (M[0] * M[4] - M[1] * M[3]) / det};
  double nv[3];
  for (int a = 0; a < 3; ++a) nv[a] = inv[3 * a] * b[0] + inv[3 * a + 1] * b[1] + inv[3 * a + 2] * b[2];
  double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
  int m = c->m++;
  for (int a = 0; a < 3; ++a) { c->out_p[3 * m + a] = mean[a]; c->out_n[3 * m + a] = nv[a] / nn; }
  c->out_k[m] = kmin; c->out_sz[m] = n;
}

static void normals_rec(NormCtx* c, int first, int last, double lo[3], double hi[3]) {  /* :105-144 */
  int count = last - first;
  if (count <= 7) { normals_leaf(c, first, last); return; }
  int dim = 0; double mv = 0.0;                                                          /* ArgMax :41-56 */
  for (int i = 0; i < 3; ++i) if (hi[i] - lo[i] > mv) { mv = hi[i] - lo[i]; dim = i; }
  int right = count / 2, left = count - right;
  kd_select(c->pts, c->indices, first, last, first + left, dim);                         /* :122-125 */
  double cut = c->pts[3 * c->indices[first + left] + dim];
  double lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
  lhi[dim] = cut; rlo[dim] = cut;
  normals_rec(c, first, first + left, lo, lhi);
  normals_rec(c, first + left, last, rlo, hi);
}

static int cmp_kept(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* returns number of surviving points M; outputs ordered by smallest source index per leaf */
int smref_calculate_normals(const double* pts, int n, double* out_pts, double* out_nrm, int* out_leaf_size) {
  if (n <= 0) return 0;
  NormCtx c;
  c.pts = pts; c.m = 0;
  c.indices = (int*)malloc(sizeof(int) * (size_t)n);
  int cap = n;
  c.out_p = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
  c.out_n = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
  c.out_k = (int*)malloc(sizeof(int) * (size_t)cap);
  c.out_sz = (int*)malloc(sizeof(int) * (size_t)cap);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i) {
    c.indices[i] = i;
    for (int d = 0; d < 3; ++d) { if (pts[3 * i + d] < lo[d]) lo[d] = pts[3 * i + d]; if (pts[3 * i + d] > hi[d]) hi[d] = pts[3 * i + d]; }
  }
  normals_rec(&c, 0, n, lo, hi);
  /* sort(indices_to_keep) :358 */
  int* order = (int*)malloc(sizeof(int) * 2 * (size_t)(c.m > 0 ? c.m : 1));
  for (int i = 0; i < c.m; ++i) { order[2 * i] = c.out_k[i]; order[2 * i + 1] = i; }
  qsort(order, c.m, 2 * sizeof(int), cmp_kept);
  for (int i = 0; i < c.m; ++i) {
    int s = order[2 * i + 1];
    for (int a = 0; a < 3; ++a) { out_pts[3 * i + a] = c.out_p[3 * s + a]; out_nrm[3 * i + a] = c.out_n[3 * s + a]; }
    if (out_leaf_size) out_leaf_size[i] = c.out_sz[s];
  }
  int m = c.m;
  free(order); free(c.indices); free(c.out_p); free(c.out_n); free(c.out_k); free(c.out_sz);
  return m;
}
