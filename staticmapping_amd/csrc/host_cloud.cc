// host_cloud.cc -- host-side mirror of the reference's target preparation step.
//
// EigenPointCloud::CalculateNormals (/root/reference/builder/data/cloud_types.cc:347-368,
// BuildNormals :105-144, leaf :73-103) is run by the CALLER of the registrator
// (builder/map_builder.cc:286,389; builder/submap.cc:161), not by Align, so this entry point is host
// code like the reference's; the device version is csrc/prep_normals.hip.  Plain C++17,
// no Eigen: kd-box split on the widest bbox axis with std::nth_element until <= 7 points, one
// surviving point (the leaf mean) + unconstrained-least-squares normal per leaf.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/smhip.h"

namespace {

constexpr int kNormalEstimationKnn = 7;   // cloud_types.cc:38

struct Leaf { int key; double p[3]; double n[3]; };

struct Builder {
  const double* pts;          // column-major 3xN == xyzxyz...
  std::vector<int> indices;
  std::vector<Leaf> leaves;

  static int ArgMaxRef(const double v[3]) {   // cloud_types.cc:41-56: starts from (index 0, value 0)
    double best = 0.0; int idx = 0;
    for (int i = 0; i < 3; ++i) if (v[i] > best) { best = v[i]; idx = i; }
    return idx;
  }

  // rank of a symmetric 3x3 matrix from its eigenvalues (Jacobi); stands in for
  // fullPivHouseholderQr().rank() at cloud_types.cc:90 (threshold eps * 3 * max pivot)
  static int Rank3(const double C[9]) {
    double A[9]; std::memcpy(A, C, sizeof(A));
    for (int sweep = 0; sweep < 50; ++sweep) {
      const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
      if (off < 1e-300) break;
      for (int p = 0; p < 3; ++p)
        for (int q = p + 1; q < 3; ++q) {
          const double apq = A[3 * p + q];
          if (std::fabs(apq) < 1e-300) continue;
          const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < 3; ++k) { const double akp = A[3 * k + p], akq = A[3 * k + q]; A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq; }
          for (int k = 0; k < 3; ++k) { const double apk = A[3 * p + k], aqk = A[3 * q + k]; A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk; }
        }
    }
    const double w[3] = {std::fabs(A[0]), std::fabs(A[4]), std::fabs(A[8])};
    const double wmax = std::max(w[0], std::max(w[1], w[2]));
    int r = 0;
    for (int i = 0; i < 3; ++i) if (w[i] > 2.220446049250313e-16 * 3 * wmax) ++r;
    return r;
  }

  void LeafNormals(int first, int last) {      // cloud_types.cc:73-103
    const int n = last - first;
    if (n <= 0) return;
    double M[9] = {0}, b[3] = {0};
    for (int i = first; i < last; ++i) {
      const double* p = pts + 3 * (size_t)indices[i];
      for (int a = 0; a < 3; ++a) { b[a] += p[a]; for (int c = 0; c < 3; ++c) M[3 * a + c] += p[a] * p[c]; }
    }
    const double mean[3] = {b[0] / n, b[1] / n, b[2] / n};
    double C[9] = {0};
    for (int i = first; i < last; ++i) {
      const double* p = pts + 3 * (size_t)indices[i];
      const double e[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
      for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] += e[a] * e[c];
    }
    if (Rank3(C) + 1 < 3) return;              // :90-92
    // normal = M^-1 * b with the cofactor inverse Eigen uses for fixed 3x3 (:94)
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double invdet = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
    const double inv[9] = {c00 * invdet, (M[2] * M[7] - M[1] * M[8]) * invdet, (M[1] * M[5] - M[2] * M[4]) * invdet,
                           c01 * invdet, (M[0] * M[8] - M[2] * M[6]) * invdet, (M[2] * M[3] - M[0] * M[5]) * invdet,
                           c02 * invdet, (M[1] * M[6] - M[0] * M[7]) * invdet, (M[0] * M[4] - M[1] * M[3]) * invdet};
    Leaf lf;
    double nn = 0;
    for (int a = 0; a < 3; ++a) { lf.n[a] = inv[3 * a] * b[0] + inv[3 * a + 1] * b[1] + inv[3 * a + 2] * b[2]; nn += lf.n[a] * lf.n[a]; }
    nn = std::sqrt(nn);
    // The reference does not check M for singularity (a plane through the sensor origin gives inf/NaN
    // here and would poison A in IcpFast); such leaves are dropped instead of kept with a NaN normal.
    if (!(nn > 0.0) || !std::isfinite(nn)) return;
    for (int a = 0; a < 3; ++a) { lf.n[a] /= nn; lf.p[a] = mean[a]; }   // :101-102
    lf.key = indices[first];                                             // :96-98: k = indices[first]
    leaves.push_back(lf);
  }

  void Build(int first, int last, const double lo[3], const double hi[3]) {   // cloud_types.cc:105-144
    const int count = last - first;
    if (count <= kNormalEstimationKnn) { LeafNormals(first, last); return; }
    const double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const int dim = ArgMaxRef(ext);
    const int right = count / 2, left = count - right;
    std::nth_element(indices.begin() + first, indices.begin() + first + left, indices.begin() + last,
                     [&](int a, int b) { return pts[3 * (size_t)a + dim] < pts[3 * (size_t)b + dim]; });   // :122-125
    const double cut = pts[3 * (size_t)indices[first + left] + dim];
    double lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
    lhi[dim] = cut; rlo[dim] = cut;
    Build(first, first + left, lo, lhi);
    Build(first + left, last, rlo, hi);
  }
};

}  // namespace

extern "C" {

// xyz: 3xN column-major doubles (EigenPointCloud::points).  out_xyz / out_normals need room for
// n points each; *n_out receives the number of surviving points (about n / 5.5).
smhip_status smhip_calculate_normals_f64(const double* xyz, int n, double* out_xyz, double* out_normals, int* n_out) {
  if (!xyz || n <= 0 || !out_xyz || !out_normals || !n_out) return SMHIP_ERR_INVALID_ARGUMENT;
  Builder bld;
  bld.pts = xyz;
  bld.indices.resize(n);
  std::iota(bld.indices.begin(), bld.indices.end(), 0);
  bld.leaves.reserve(n / 4 + 4);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], xyz[3 * (size_t)i + d]); hi[d] = std::max(hi[d], xyz[3 * (size_t)i + d]); }
  bld.Build(0, n, lo, hi);
  std::sort(bld.leaves.begin(), bld.leaves.end(), [](const Leaf& a, const Leaf& b) { return a.key < b.key; });   // :358
  int m = 0;
  for (const Leaf& lf : bld.leaves) {
    for (int a = 0; a < 3; ++a) { out_xyz[3 * (size_t)m + a] = lf.p[a]; out_normals[3 * (size_t)m + a] = lf.n[a]; }
    ++m;
  }
  *n_out = m;
  return SMHIP_OK;
}

}  // extern "C"
