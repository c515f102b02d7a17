/* smref_ndt.c -- dependency-free C restatement of registrators::Ndt (pclomp NDT), the timed CPU baseline of
 * BASELINE config #3 and a second, independent check of oracle/ndt.py.  TEST ORACLE / CPU BASELINE ONLY; never linked
 * into the product.  PARITY UNPINNED (no reference test pins these results; the reference cannot be built here).
 *
 * Restates (paths relative to /root/reference/registrators):
 *   ndt.cc:29-64                                          wrapper: resolution 1.0, 6 threads, KDTREE search, fitness
 *   pclomp/voxel_grid_covariance_omp_impl.hpp:49-370      applyFilter
 *   pclomp/voxel_grid_covariance_omp.h:92-106,204-205,470-499   Leaf (cov_ = I), >= 6 points, radiusSearch
 *   pclomp/ndt_omp_impl.hpp:81-171                        computeTransformation (Newton + More-Thuente driver)
 *   pclomp/ndt_omp_impl.hpp:180-284                       computeDerivatives (OpenMP over points, schedule(guided, 8))
 *   pclomp/ndt_omp_impl.hpp:288-438, 483-535              angle / point derivatives, updateDerivatives (float inner math)
 *   pclomp/ndt_omp_impl.hpp:633-916                       updateIntervalMT / trialValueSelectionMT / computeStepLengthMT
 * Third-party behaviour restated: FLANN radius search over the voxel centroids (here: the 27 voxels around the point's
 * voxel, filtered by the float squared centroid distance -- the same set, because a centroid lies inside its voxel and
 * radius == resolution), pcl::transformPointCloud (float 4x4), pcl::Registration::getFitnessScore (mean squared 1-NN
 * distance to the raw target), Eigen JacobiSVD::solve (pseudo-inverse), Eigen eulerAngles(0, 1, 2).
 * Matrices crossing this C API are ROW-major 4x4 doubles. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "smref_internal.h"

typedef struct {
  int nvox;
  long* key;       /* linear voxel index (:223) */
  double* mean;    /* [nvox][3] */
  double* icov;    /* [nvox][9] */
  float* cent;     /* [nvox][3] */
  int* valid;      /* eigen check passed (:337-341) */
  int* npts;
  long min_b[3], div_b[3];
  float inv;
  float resolution;
  /* open-addressing hash key -> voxel */
  long* hkey; int* hval; long hmask;
} NdtGrid;

static int cmp_long_pair(const void* a, const void* b) {
  const long* x = (const long*)a; const long* y = (const long*)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

static void inv3(const double* m, double* o) {           /* Eigen fixed 3x3 inverse = cofactors / det */
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

static int grid_lookup(const NdtGrid* g, long key) {
  long h = (long)(((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 20) & g->hmask;
  while (g->hkey[h] != -1) { if (g->hkey[h] == key) return g->hval[h]; h = (h + 1) & g->hmask; }
  return -1;
}

void smref_ndt_grid_free(NdtGrid* g) {
  if (!g) return;
  free(g->key); free(g->mean); free(g->icov); free(g->cent); free(g->valid); free(g->npts); free(g->hkey); free(g->hval); free(g);
}

/* VoxelGridCovariance::applyFilter.  tgt: n rows of 3 floats. */
NdtGrid* smref_ndt_grid_build(const float* tgt, int n, float resolution, int min_points, double eig_mult) {
  NdtGrid* g = (NdtGrid*)calloc(1, sizeof(NdtGrid));
  g->resolution = resolution;
  const float inv = 1.0f / resolution;                                   /* inverse_leaf_size_ */
  g->inv = inv;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int nf = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = tgt + 3 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    ++nf;
    for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }   /* getMinMax3D :71 */
  }
  if (nf == 0) { g->hmask = 0; g->hkey = (long*)malloc(sizeof(long)); g->hkey[0] = -1; g->hval = (int*)calloc(1, sizeof(int)); return g; }
  long max_b[3];
  for (int d = 0; d < 3; ++d) { g->min_b[d] = (long)floorf(mn[d] * inv); max_b[d] = (long)floorf(mx[d] * inv); g->div_b[d] = max_b[d] - g->min_b[d] + 1; }  /* :87-95 */
  long* pairs = (long*)malloc(sizeof(long) * 2 * (size_t)nf);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = tgt + 3 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    long ijk[3];
    for (int d = 0; d < 3; ++d) ijk[d] = (long)(floorf(p[d] * inv) - (float)g->min_b[d]);              /* :218-220 */
    pairs[2 * m] = ijk[0] + ijk[1] * g->div_b[0] + ijk[2] * g->div_b[0] * g->div_b[1];               /* :223 */
    pairs[2 * m + 1] = i; ++m;
  }
  qsort(pairs, (size_t)m, 2 * sizeof(long), cmp_long_pair);              /* std::map order, points in arrival order */
  int cap = 0;
  for (int s = 0; s < m;) { int e = s; while (e < m && pairs[2 * e] == pairs[2 * s]) ++e; if (e - s >= min_points) ++cap; s = e; }
  g->key = (long*)malloc(sizeof(long) * (size_t)(cap + 1)); g->mean = (double*)malloc(sizeof(double) * 3 * (size_t)(cap + 1));
  g->icov = (double*)calloc(9 * (size_t)(cap + 1), sizeof(double)); g->cent = (float*)malloc(sizeof(float) * 3 * (size_t)(cap + 1));
  g->valid = (int*)malloc(sizeof(int) * (size_t)(cap + 1)); g->npts = (int*)malloc(sizeof(int) * (size_t)(cap + 1));
  int v = 0;
  for (int s = 0; s < m;) {
    int e = s; while (e < m && pairs[2 * e] == pairs[2 * s]) ++e;
    const int cnt = e - s;
    if (cnt >= min_points) {                                                                       /* :297 */
      double sum[3] = {0, 0, 0}, cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};                            /* Leaf(): cov_ = Identity */
      float cs[3] = {0, 0, 0};
      for (int k = s; k < e; ++k) {
        const float* p = tgt + 3 * (size_t)pairs[2 * k + 1];
        const double q[3] = {p[0], p[1], p[2]};
        for (int a = 0; a < 3; ++a) { sum[a] += q[a]; cs[a] += p[a]; for (int b = 0; b < 3; ++b) cov[3 * a + b] += q[a] * q[b]; }   /* :233-241 */
      }
      double mean[3] = {sum[0] / cnt, sum[1] / cnt, sum[2] / cnt};                                 /* :293 */
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          cov[3 * a + b] = ((cov[3 * a + b] - 2.0 * sum[a] * mean[b]) / cnt + mean[a] * mean[b]) * ((cnt - 1.0) / cnt);   /* :329-330 */
      double E[9], V[9], w[3];
      memcpy(E, cov, sizeof(E));
      jacobi_eig(3, E, V, w);
      int o[3] = {0, 1, 2};                                                                         /* ascending, like SelfAdjointEigenSolver */
      for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (w[o[b]] < w[o[a]]) { int t = o[a]; o[a] = o[b]; o[b] = t; }
      double ws[3] = {w[o[0]], w[o[1]], w[o[2]]};
      int ok = 1;
      double* ic = g->icov + 9 * (size_t)v;
      if (ws[0] < 0 || ws[1] < 0 || ws[2] <= 0) ok = 0;                                             /* :337-341 */
      else {
        const double mm = eig_mult * ws[2];                                                         /* :345 */
        if (ws[0] < mm) {                                                                           /* :346-356 */
          ws[0] = mm; if (ws[1] < mm) ws[1] = mm;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
              double acc = 0;
              for (int k = 0; k < 3; ++k) acc += V[3 * a + o[k]] * ws[k] * V[3 * b + o[k]];        /* V diag(w) V^-1, V orthonormal */
              cov[3 * a + b] = acc;
            }
        }
        inv3(cov, ic);                                                                              /* :359 */
        for (int k = 0; k < 9; ++k) if (!isfinite(ic[k])) ok = 0;
        if (!ok) memset(ic, 0, 9 * sizeof(double));
      }
      g->key[v] = pairs[2 * s]; g->valid[v] = ok; g->npts[v] = cnt;
      for (int a = 0; a < 3; ++a) { g->mean[3 * v + a] = mean[a]; g->cent[3 * v + a] = cs[a] / (float)cnt; }   /* :289 */
      ++v;
    }
    s = e;
  }
  g->nvox = v;
  free(pairs);
  long hcap = 8; while (hcap < 2L * (v + 1)) hcap <<= 1;
  g->hmask = hcap - 1;
  g->hkey = (long*)malloc(sizeof(long) * (size_t)hcap); g->hval = (int*)malloc(sizeof(int) * (size_t)hcap);
  for (long k = 0; k < hcap; ++k) g->hkey[k] = -1;
  for (int k = 0; k < v; ++k) {
    long h = (long)(((uint64_t)g->key[k] * 0x9E3779B97F4A7C15ull) >> 20) & g->hmask;
    while (g->hkey[h] != -1) h = (h + 1) & g->hmask;
    g->hkey[h] = g->key[k]; g->hval[h] = k;
  }
  return g;
}

int smref_ndt_grid_size(const NdtGrid* g) { return g->nvox; }
void smref_ndt_grid_get(const NdtGrid* g, long* key, double* mean, double* icov, float* cent, int* valid) {
  memcpy(key, g->key, sizeof(long) * (size_t)g->nvox); memcpy(mean, g->mean, sizeof(double) * 3 * (size_t)g->nvox);
  memcpy(icov, g->icov, sizeof(double) * 9 * (size_t)g->nvox); memcpy(cent, g->cent, sizeof(float) * 3 * (size_t)g->nvox);
  memcpy(valid, g->valid, sizeof(int) * (size_t)g->nvox);
}

static void gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2) {     /* ndt_omp_impl.hpp:86-93 */
  const double c1 = 10.0 * (1 - outlier_ratio), c2 = outlier_ratio / (resolution * resolution * resolution);
  const double d3 = -log(c2);
  *d1 = -log(c1 + c2) - d3;
  *d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / *d1);
}

static void angle_derivatives(const double p[6], float j[8][3], float h[15][3]) {                   /* :288-393 */
  double cx, sx, cy, sy, cz, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1; sx = 0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1; sy = 0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1; sz = 0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  const double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                          {-sy * cz, sy * sz, cy}, {sx * cy * cz, -sx * cy * sz, sx * sy}, {-cx * cy * cz, cx * cy * sz, -cx * sy},
                          {-cy * sz, -cy * cz, 0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0}, {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  const double H[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                           {cx * cy * cz, -cx * cy * sz, cx * sy}, {sx * cy * cz, -sx * cy * sz, sx * sy},
                           {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                           {-cy * cz, cy * sz, sy}, {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy},
                           {sy * sz, sy * cz, 0}, {-sx * cy * sz, -sx * cy * cz, 0}, {cx * cy * sz, cx * cy * cz, 0},
                           {-cy * cz, cy * sz, 0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) j[r][c] = (float)J[r][c];
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) h[r][c] = (float)H[r][c];
}

/* computeDerivatives: src / trans = n rows of 3 floats.  hess row-major 6x6.  Returns the number of (point, voxel) pairs used. */
long smref_ndt_compute_derivatives(const NdtGrid* g, const float* src, const float* trans, int n, const double p[6],
                                   double outlier_ratio, int compute_hessian, int nthreads,
                                   double* score_out, double* grad, double* hess) {
  double d1, d2;
  gauss_constants(g->resolution, outlier_ratio, &d1, &d2);
  float j_ang[8][3], h_ang[15][3];
  angle_derivatives(p, j_ang, h_ang);
  if (nthreads < 1) nthreads = 1;
  double* part = (double*)calloc((size_t)nthreads * 44, sizeof(double));        /* score, g[6], H[36], pairs */
  const float gd2 = (float)d2, r2 = g->resolution * g->resolution;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(guided, 8)
#endif
  for (int i = 0; i < n; ++i) {
#ifdef _OPENMP
    double* acc = part + 44 * (size_t)omp_get_thread_num();
#else
    double* acc = part;
#endif
    const float* x = src + 3 * (size_t)i; const float* xt = trans + 3 * (size_t)i;
    if (!isfinite(xt[0]) || !isfinite(xt[1]) || !isfinite(xt[2])) continue;
    long c[3];
    for (int d = 0; d < 3; ++d) c[d] = (long)(floorf(xt[d] * g->inv) - (float)g->min_b[d]);
    float xj[8], xh[15];
    int have_j = 0;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const long ix = c[0] + dx, iy = c[1] + dy, iz = c[2] + dz;
      if (ix < 0 || iy < 0 || iz < 0 || ix >= g->div_b[0] || iy >= g->div_b[1] || iz >= g->div_b[2]) continue;
      const int v = grid_lookup(g, ix + iy * g->div_b[0] + iz * g->div_b[0] * g->div_b[1]);
      if (v < 0) continue;
      const float* ce = g->cent + 3 * (size_t)v;
      const float ex = xt[0] - ce[0], ey = xt[1] - ce[1], ez = xt[2] - ce[2];
      if (ex * ex + ey * ey + ez * ez > r2) continue;                               /* radiusSearch, .h:470-499 */
      if (!have_j) {                                                                /* computePointDerivatives :397-438 */
        for (int r = 0; r < 8; ++r) xj[r] = x[0] * j_ang[r][0] + x[1] * j_ang[r][1] + x[2] * j_ang[r][2];
        if (compute_hessian) for (int r = 0; r < 15; ++r) xh[r] = x[0] * h_ang[r][0] + x[1] * h_ang[r][1] + x[2] * h_ang[r][2];
        have_j = 1;
      }
      /* updateDerivatives :483-535 */
      float J[3][6] = {{1, 0, 0, 0, xj[2], xj[5]}, {0, 1, 0, xj[0], xj[3], xj[6]}, {0, 0, 1, xj[1], xj[4], xj[7]}};
      const double* mu = g->mean + 3 * (size_t)v; const double* ic = g->icov + 9 * (size_t)v;
      const float t[3] = {(float)((double)xt[0] - mu[0]), (float)((double)xt[1] - mu[1]), (float)((double)xt[2] - mu[2])};   /* :253, :490 */
      float C[3][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a][b] = (float)ic[3 * a + b];
      float xC[3];
      for (int b = 0; b < 3; ++b) xC[b] = t[0] * C[0][b] + t[1] * C[1][b] + t[2] * C[2][b];
      const float q = xC[0] * t[0] + xC[1] * t[1] + xC[2] * t[2];
      const float e = expf(-gd2 * q * 0.5f);                                        /* :497 */
      const float score_inc = (float)(-d1 * (double)e);                             /* :499 */
      float e2 = gd2 * e;                                                           /* :501 */
      if (e2 > 1 || e2 < 0 || e2 != e2) continue;                                   /* :504-505 */
      e2 = (float)(d1 * (double)e2);                                                /* :508 */
      float CJ[3][6], xCJ[6];
      for (int a = 0; a < 3; ++a) for (int l = 0; l < 6; ++l) CJ[a][l] = C[a][0] * J[0][l] + C[a][1] * J[1][l] + C[a][2] * J[2][l];   /* :510 */
      for (int l = 0; l < 6; ++l) xCJ[l] = t[0] * CJ[0][l] + t[1] * CJ[1][l] + t[2] * CJ[2][l];                                      /* :511 */
      acc[0] += (double)score_inc;
      for (int l = 0; l < 6; ++l) acc[1 + l] += (double)(e2 * xCJ[l]);             /* :513 */
      acc[43] += 1.0;
      if (compute_hessian) {
        /* point_hessian_ blocks (:418-437): rows (i - 3), columns (j - 3) of 3-vectors (y, z from xh; x for the lower ones) */
        const float a_[3] = {0, xh[0], xh[1]}, b_[3] = {0, xh[2], xh[3]}, c_[3] = {0, xh[4], xh[5]};
        const float d_[3] = {xh[6], xh[7], xh[8]}, e_[3] = {xh[9], xh[10], xh[11]}, f_[3] = {xh[12], xh[13], xh[14]};
        const float* PH[3][3] = {{a_, b_, c_}, {b_, d_, e_}, {c_, e_, f_}};
        for (int ii = 0; ii < 6; ++ii)
          for (int jj = 0; jj < 6; ++jj) {
            float xCH = 0.f;
            if (ii >= 3 && jj >= 3) { const float* ph = PH[ii - 3][jj - 3]; xCH = xC[0] * ph[0] + xC[1] * ph[1] + xC[2] * ph[2]; }   /* :523 */
            const float JCJ = J[0][jj] * CJ[0][ii] + J[1][jj] * CJ[1][ii] + J[2][jj] * CJ[2][ii];                                 /* :517 */
            const float term = -gd2 * xCJ[ii] * xCJ[jj] + xCH + JCJ;                                                              /* :527-529 */
            acc[7 + 6 * ii + jj] += (double)(e2 * term);
          }
      }
    }
  }
  double sc = 0; long pairs = 0;
  for (int k = 0; k < 6; ++k) grad[k] = 0;
  if (hess) for (int k = 0; k < 36; ++k) hess[k] = 0;
  for (int t = 0; t < nthreads; ++t) {
    const double* a = part + 44 * (size_t)t;
    sc += a[0]; pairs += (long)a[43];
    for (int k = 0; k < 6; ++k) grad[k] += a[1 + k];
    if (hess && compute_hessian) for (int k = 0; k < 36; ++k) hess[k] += a[7 + k];
  }
  free(part);
  *score_out = sc;
  return pairs;
}

static void pose_to_matrix_f32(const double p[6], float T[16]) {                    /* :146-149, 808-811: Translation * Rx * Ry * Rz, float */
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = cosf(a), sa = sinf(a), cb = cosf(b), sb = sinf(b), cc = cosf(c), sc = sinf(c);
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca}, Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb}, Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float M[9], R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[3 * i + j] = Rx[3 * i] * Ry[j] + Rx[3 * i + 1] * Ry[3 + j] + Rx[3 * i + 2] * Ry[6 + j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = M[3 * i] * Rz[j] + M[3 * i + 1] * Rz[3 + j] + M[3 * i + 2] * Rz[6 + j];
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = (float)p[i]; }
}

static void transform_cloud_f32(const float* x, int n, const float T[16], float* out, int nthreads) {   /* pcl::transformPointCloud */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < 3; ++r)
      out[3 * (size_t)i + r] = x[3 * (size_t)i] * T[4 * r] + x[3 * (size_t)i + 1] * T[4 * r + 1] + x[3 * (size_t)i + 2] * T[4 * r + 2] + T[4 * r + 3];
  (void)nthreads;
}

static void euler_xyz(const double R[9], double out[3]) {                           /* Eigen eulerAngles(0, 1, 2) */
  double r0 = atan2(R[5], R[8]);
  const double c2 = hypot(R[0], R[1]);
  double r1;
  if (r0 > 0.0) { r0 -= M_PI; r1 = atan2(-R[2], -c2); } else r1 = atan2(-R[2], c2);
  const double s1 = sin(r0), c1 = cos(r0);
  const double r2 = atan2(s1 * R[6] - c1 * R[3], c1 * R[4] - s1 * R[7]);
  out[0] = -r0; out[1] = -r1; out[2] = -r2;
}

static void svd_solve6(const double* H, const double* b, double* x) {               /* JacobiSVD(H).solve(b), H symmetric up to rounding */
  double S[36], V[36], w[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) S[6 * i + j] = 0.5 * (H[6 * i + j] + H[6 * j + i]);
  jacobi_eig(6, S, V, w);
  double wmax = 0;
  for (int i = 0; i < 6; ++i) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
  const double thr = 2.220446049250313e-16 * 6 * wmax;
  for (int i = 0; i < 6; ++i) x[i] = 0;
  for (int k = 0; k < 6; ++k) {
    if (fabs(w[k]) <= thr) continue;
    double c = 0;
    for (int i = 0; i < 6; ++i) c += V[6 * i + k] * b[i];
    c /= w[k];
    for (int i = 0; i < 6; ++i) x[i] += c * V[6 * i + k];
  }
}

/* More-Thuente helpers, ndt_omp_impl.hpp:633-753 */
static int update_interval(double* a_l, double* f_l, double* g_l, double* a_u, double* f_u, double* g_u, double a_t, double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  return 1;
}
static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double nxt = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    const double lim = a_t + 0.66 * (a_u - a_t);
    return a_t > a_l ? (lim < nxt ? lim : nxt) : (lim > nxt ? lim : nxt);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

/* Ndt::Align (ndt.cc:38-64).  src: ns rows of 3 floats, tgt: nt rows of 3 floats; guess / result row-major 4x4 doubles.
 * nthreads_deriv: computeDerivatives threads (the reference: 6, ndt.cc:32); nthreads_other: transform / fitness threads
 * (the reference: 1).  block_times[5]: applyFilter, computeDerivatives (summed), transformPointCloud (summed),
 * getFitnessScore (kd-tree build + queries), whole Align. */
int smref_ndt_align(const float* src, int ns, const float* tgt, int nt, const double* guess, float resolution,
                    double step_size, double outlier_ratio, double trans_eps, int max_iterations,
                    int nthreads_deriv, int nthreads_other, int with_fitness,
                    double* result, double* fitness, int* iterations, int* derivative_calls, double* trans_probability,
                    double* mean_neighbours, int* n_voxels, double* block_times) {
  const double t_begin = now_s();
  double t_grid, t_der = 0, t_tf = 0, t_fit = 0, t0;
  t0 = now_s();
  NdtGrid* grid = smref_ndt_grid_build(tgt, nt, resolution, 6, 0.01);
  t_grid = now_s() - t0;
  float G[16], fin[16];
  for (int i = 0; i < 16; ++i) { G[i] = (float)guess[i]; fin[i] = G[i]; }                                  /* guess.cast<float>(), ndt.cc:58 */
  float* trans = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
  int is_identity = 1;
  for (int i = 0; i < 16; ++i) if (G[i] != ((i % 5 == 0) ? 1.f : 0.f)) is_identity = 0;
  t0 = now_s();
  if (is_identity) memcpy(trans, src, sizeof(float) * 3 * (size_t)ns); else transform_cloud_f32(src, ns, G, trans, nthreads_other);   /* :95-101 */
  t_tf += now_s() - t0;
  double p[6], R[9], eul[3];
  for (int i = 0; i < 3; ++i) { p[i] = (double)fin[4 * i + 3]; for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)fin[4 * i + j]; }   /* :107-111 */
  euler_xyz(R, eul);
  for (int i = 0; i < 3; ++i) p[3 + i] = (double)(float)eul[i];
  int calls = 0, it = 0, converged = 0;
  double score, g[6], H[36];
  long pairs;
  t0 = now_s();
  pairs = smref_ndt_compute_derivatives(grid, src, trans, ns, p, outlier_ratio, 1, nthreads_deriv, &score, g, H); ++calls;   /* :119 */
  t_der += now_s() - t0;
  while (!converged) {                                                                                     /* :121 */
    double mg[6], dp[6], step_dir[6];
    for (int i = 0; i < 6; ++i) mg[i] = -g[i];
    svd_solve6(H, mg, dp);                                                                                 /* :127-129 */
    double dp_norm = 0;
    for (int i = 0; i < 6; ++i) dp_norm += dp[i] * dp[i];
    dp_norm = sqrt(dp_norm);
    if (dp_norm == 0 || dp_norm != dp_norm) break;                                                         /* :134-139 */
    for (int i = 0; i < 6; ++i) step_dir[i] = dp[i] / dp_norm;                                             /* :141 */
    /* computeStepLengthMT(p, step_dir, dp_norm, step_size, trans_eps / 2, ...)  :757-916 */
    const double step_init = dp_norm, step_max = step_size, step_min = trans_eps / 2;
    const double phi_0 = -score;
    double d_phi_0 = 0;
    for (int i = 0; i < 6; ++i) d_phi_0 -= g[i] * step_dir[i];
    double a_t = 0.0;
    int skip = 0;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) skip = 1;
      else { d_phi_0 = -d_phi_0; for (int i = 0; i < 6; ++i) step_dir[i] = -step_dir[i]; }
    }
    if (!skip) {
      const double mu = 1e-4, nu = 0.9;
      double a_l = 0, a_u = 0, f_l = 0 /* psi(0) */, g_l = d_phi_0 - mu * d_phi_0, f_u = f_l, g_u = g_l;
      int interval_converged = (step_max - step_min) > 0;                                                   /* :795 (sic) */
      int open_interval = 1;
      a_t = step_init < step_max ? step_init : step_max; if (a_t < step_min) a_t = step_min;                /* :797-799 */
      double x_t[6];
      for (int i = 0; i < 6; ++i) x_t[i] = p[i] + step_dir[i] * a_t;
      pose_to_matrix_f32(x_t, fin);                                                                         /* :803-806 */
      t0 = now_s(); transform_cloud_f32(src, ns, fin, trans, nthreads_other); t_tf += now_s() - t0;         /* :809 */
      t0 = now_s();
      pairs = smref_ndt_compute_derivatives(grid, src, trans, ns, x_t, outlier_ratio, 1, nthreads_deriv, &score, g, H); ++calls;   /* :813 */
      t_der += now_s() - t0;
      double phi_t = -score, d_phi_t = 0;
      for (int i = 0; i < 6; ++i) d_phi_t -= g[i] * step_dir[i];
      double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
      int step_iterations = 0;
      while (!interval_converged && step_iterations < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                            : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        if (a_t > step_max) a_t = step_max;
        if (a_t < step_min) a_t = step_min;
        for (int i = 0; i < 6; ++i) x_t[i] = p[i] + step_dir[i] * a_t;
        pose_to_matrix_f32(x_t, fin);
        t0 = now_s(); transform_cloud_f32(src, ns, fin, trans, nthreads_other); t_tf += now_s() - t0;
        double Hd[36];
        t0 = now_s();
        pairs = smref_ndt_compute_derivatives(grid, src, trans, ns, x_t, outlier_ratio, 0, nthreads_deriv, &score, g, Hd); ++calls;
        t_der += now_s() - t0;
        phi_t = -score; d_phi_t = 0;
        for (int i = 0; i < 6; ++i) d_phi_t -= g[i] * step_dir[i];
        psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t; d_psi_t = d_phi_t - mu * d_phi_0;
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          open_interval = 0;
          f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
          f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        interval_converged = open_interval ? update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t)
                                           : update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        ++step_iterations;
      }
      if (step_iterations) {                                                                                /* :912-913 */
        double gd[6], sd;
        t0 = now_s();
        smref_ndt_compute_derivatives(grid, src, trans, ns, x_t, outlier_ratio, 1, nthreads_deriv, &sd, gd, H); ++calls;
        t_der += now_s() - t0;
      }
    }
    dp_norm = a_t;                                                                                          /* :142 */
    for (int i = 0; i < 6; ++i) p[i] += step_dir[i] * dp_norm;                                              /* :143, :152 */
    if (it > max_iterations || (it && fabs(dp_norm) < trans_eps)) converged = 1;                            /* :158-162 */
    ++it;                                                                                                   /* :164 */
  }
  for (int i = 0; i < 16; ++i) result[i] = (double)fin[i];
  *iterations = it; *derivative_calls = calls; *trans_probability = score / ns;
  if (mean_neighbours) *mean_neighbours = (double)pairs / ns;
  if (n_voxels) *n_voxels = grid->nvox;
  *fitness = 0;
  if (with_fitness) {                                                                                       /* pcl::Registration::getFitnessScore, ndt.cc:60 */
    t0 = now_s();
    transform_cloud_f32(src, ns, fin, trans, nthreads_other);
    double* T = (double*)malloc(sizeof(double) * 3 * (size_t)nt);
    for (size_t k = 0; k < 3 * (size_t)nt; ++k) T[k] = (double)tgt[k];
    KdTree* tree = kd_build(T, nt);
    double acc = 0; long cnt = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads_other > 0 ? nthreads_other : 1) schedule(guided, 32) reduction(+ : acc, cnt)
#endif
    for (int i = 0; i < ns; ++i) {
      const double q[3] = {(double)trans[3 * (size_t)i], (double)trans[3 * (size_t)i + 1], (double)trans[3 * (size_t)i + 2]};
      int id; double d2;
      kd_nn(tree, q, &id, &d2);
      const float df = (float)sqrt(d2);
      acc += (double)(df * df); ++cnt;
    }
    *fitness = cnt ? acc / (double)cnt : 0.0;
    kd_free(tree); free(T);
    t_fit = now_s() - t0;
  }
  if (block_times) { block_times[0] = t_grid; block_times[1] = t_der; block_times[2] = t_tf; block_times[3] = t_fit; block_times[4] = now_s() - t_begin; }
  free(trans);
  smref_ndt_grid_free(grid);
  return 0;
}
