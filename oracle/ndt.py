"""numpy/scipy restatement of `registrators::Ndt` (pclomp NDT) -- TEST ORACLE, PARITY UNPINNED
(see oracle/__init__.py).  Paths are relative to /root/reference/registrators.

Restates
  ndt.cc:29-64                                   wrapper: resolution 1.0, KDTREE neighbourhood, fitness score
  pclomp/voxel_grid_covariance_omp_impl.hpp:49-370   applyFilter (voxel means / covariances / inverse covariances)
  pclomp/voxel_grid_covariance_omp.h:92-106,204-205,470-499   Leaf ctor (cov_ = I), min 6 points, radiusSearch
  pclomp/ndt_omp_impl.hpp:47-76, 81-171           constants, computeTransformation
  pclomp/ndt_omp_impl.hpp:180-284                 computeDerivatives
  pclomp/ndt_omp_impl.hpp:288-438                 computeAngleDerivatives / computePointDerivatives (float)
  pclomp/ndt_omp_impl.hpp:483-535                 updateDerivatives (float inner math, double accumulation)
  pclomp/ndt_omp_impl.hpp:757-916                 computeStepLengthMT (incl. the `(step_max - step_min) > 0`
                                                  initialisation that makes the More-Thuente loop a no-op)
Third-party pieces restated from their documented behaviour: pcl::Registration::align /
getFitnessScore (mean squared 1-NN distance source->raw target), pcl::transformPointCloud (float
4x4), FLANN radius search (all centroids with squared distance <= r^2), Eigen JacobiSVD::solve,
Eigen eulerAngles(0,1,2).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

F = np.float32


# --------------------------------------------------------------------------------------------
# VoxelGridCovariance::applyFilter
# --------------------------------------------------------------------------------------------
class VoxelGrid:
    """Searchable voxels (n >= 6) of the target: centroid (f32), mean (f64), icov (f64), valid flag."""

    def __init__(self, target_f32: np.ndarray, resolution: float = 1.0, min_points: int = 6,
                 eig_mult: float = 0.01):
        pts = np.asarray(target_f32, dtype=F)[:, :3]
        finite = np.isfinite(pts).all(axis=1)
        pts = pts[finite]
        inv = F(1.0) / F(resolution)                                   # inverse_leaf_size_ (float)
        min_p, max_p = pts.min(axis=0), pts.max(axis=0)                # getMinMax3D, :71
        self.min_b = np.floor(min_p * inv).astype(np.int64)            # :87-92
        self.max_b = np.floor(max_p * inv).astype(np.int64)
        self.div_b = self.max_b - self.min_b + 1                       # :95
        self.inv = inv
        self.resolution = resolution
        ijk = (np.floor(pts * inv) - self.min_b.astype(F)).astype(np.int64)      # :218-220 (float arithmetic)
        idx = ijk[:, 0] + ijk[:, 1] * self.div_b[0] + ijk[:, 2] * self.div_b[0] * self.div_b[1]   # :223
        order = np.argsort(idx, kind="stable")
        idx_s, p_s = idx[order], pts[order].astype(np.float64)
        uniq, start, count = np.unique(idx_s, return_index=True, return_counts=True)
        means, icovs, cents, valid, keys = [], [], [], [], []
        for u, s, n in zip(uniq, start, count):
            if n < min_points:                                         # :297
                continue
            p = p_s[s:s + n]
            pt_sum = p.sum(axis=0)                                     # leaf.mean_ accumulated, :233
            cov_acc = np.eye(3) + p.T @ p                              # Leaf ctor cov_ = I (.h:101) + sum p p^T, :235
            mean = pt_sum / n                                          # :293
            cent = (pts[order][s:s + n].sum(axis=0, dtype=F) / F(n)).astype(F)     # centroid accumulated in float, :241,:289
            cov = (cov_acc - 2.0 * np.outer(pt_sum, mean)) / n + np.outer(mean, mean)    # :329
            cov *= (n - 1.0) / n                                       # :330
            w, V = np.linalg.eigh(cov)                                 # SelfAdjointEigenSolver, ascending, :333-335
            ok = True
            icov = np.zeros((3, 3))
            if w[0] < 0 or w[1] < 0 or w[2] <= 0:                      # :337-341: stays searchable with icov = 0
                ok = False
            else:
                m = eig_mult * w[2]                                    # :345
                if w[0] < m:                                           # :346-356
                    w = w.copy()
                    w[0] = m
                    if w[1] < m:
                        w[1] = m
                    cov = V @ np.diag(w) @ np.linalg.inv(V)
                icov = np.linalg.inv(cov)                              # :359
                if not np.isfinite(icov).all():                        # :360-364
                    ok = False
                    icov = np.zeros((3, 3))
            means.append(mean); icovs.append(icov); cents.append(cent); valid.append(ok); keys.append(u)
        self.mean = np.asarray(means).reshape(-1, 3)
        self.icov = np.asarray(icovs).reshape(-1, 3, 3)
        self.centroid = np.asarray(cents, dtype=F).reshape(-1, 3)
        self.valid = np.asarray(valid, dtype=bool)
        self.key = np.asarray(keys, dtype=np.int64)
        self.tree = cKDTree(self.centroid.astype(np.float64)) if len(self.centroid) else None

    def radius_pairs(self, x_trans_f32: np.ndarray):
        """(point index, voxel index) for every voxel centroid within `resolution` of the point
        (voxel_grid_covariance_omp.h:470-499 via FLANN: squared float distance <= r^2)."""
        if self.tree is None:
            return np.zeros(0, np.int64), np.zeros(0, np.int64)
        x = np.asarray(x_trans_f32, dtype=F)
        lists = self.tree.query_ball_point(x.astype(np.float64), r=self.resolution * (1 + 1e-6))
        pi = np.repeat(np.arange(len(lists)), [len(l) for l in lists])
        vi = np.fromiter((v for l in lists for v in l), dtype=np.int64, count=len(pi))
        if len(pi) == 0:
            return pi, vi
        d = x[pi] - self.centroid[vi]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F)
        keep = d2 <= F(self.resolution) * F(self.resolution)
        return pi[keep], vi[keep]


# --------------------------------------------------------------------------------------------
# derivatives
# --------------------------------------------------------------------------------------------
def gauss_constants(resolution=1.0, outlier_ratio=0.55):
    """ndt_omp_impl.hpp:86-93."""
    c1 = 10.0 * (1 - outlier_ratio)
    c2 = outlier_ratio / resolution ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2, d3


def angle_derivatives(p, real=None):
    """ndt_omp_impl.hpp:288-393: j_ang (8x4 float) and h_ang (16x4 float, 15 rows used); `real`=float64 for
    stock PCL (double rows)."""
    F = np.float32 if real is None else real
    def cs(a):
        return (1.0, 0.0) if abs(a) < 10e-5 else (np.cos(a), np.sin(a))
    cx, sx = cs(p[3]); cy, sy = cs(p[4]); cz, sz = cs(p[5])
    j = np.zeros((8, 4), dtype=F)
    j[0, :3] = (-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)
    j[1, :3] = (cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)
    j[2, :3] = (-sy * cz), sy * sz, cy
    j[3, :3] = sx * cy * cz, (-sx * cy * sz), sx * sy
    j[4, :3] = (-cx * cy * cz), cx * cy * sz, (-cx * sy)
    j[5, :3] = (-cy * sz), (-cy * cz), 0
    j[6, :3] = (cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0
    j[7, :3] = (sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0
    h = np.zeros((16, 4), dtype=F)
    h[0, :3] = (-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy
    h[1, :3] = (-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)
    h[2, :3] = (cx * cy * cz), (-cx * cy * sz), (cx * sy)
    h[3, :3] = (sx * cy * cz), (-sx * cy * sz), (sx * sy)
    h[4, :3] = (-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0
    h[5, :3] = (cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0
    h[6, :3] = (-cy * cz), (cy * sz), (sy)
    h[7, :3] = (-sx * sy * cz), (sx * sy * sz), (sx * cy)
    h[8, :3] = (cx * sy * cz), (-cx * sy * sz), (-cx * cy)
    h[9, :3] = (sy * sz), (sy * cz), 0
    h[10, :3] = (-sx * cy * sz), (-sx * cy * cz), 0
    h[11, :3] = (cx * cy * sz), (cx * cy * cz), 0
    h[12, :3] = (-cy * cz), (cy * sz), 0
    h[13, :3] = (-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0
    h[14, :3] = (-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0
    return j, h


def compute_derivatives(grid: VoxelGrid, src_f32, trans_f32, p, d1, d2, compute_hessian=True, pairs=None, real=None):
    """ndt_omp_impl.hpp:180-284 + 397-438 + 483-535.  Returns (score, gradient[6], hessian[6,6], n_pairs).
    `pairs` = (point idx, voxel idx) freezes the neighbourhoods (tests only: the score itself jumps
    whenever a point crosses a radius-search boundary, so finite differences need a fixed set).
    `real` = arithmetic type of the per-neighbour math: float32 restates pclomp (the default), float64
    restates stock pcl::NormalDistributionsTransform (PCL 1.8.1 ndt.hpp, same formulas in double) which
    registrators/ndt_gicp.cc uses."""
    F = np.float32 if real is None else real
    x = np.asarray(src_f32, dtype=np.float32)[:, :3].astype(F)
    xt = np.asarray(trans_f32, dtype=np.float32)[:, :3].astype(F)
    j_ang, h_ang = angle_derivatives(p, F)
    pi, vi = grid.radius_pairs(xt) if pairs is None else pairs
    g = np.zeros(6)
    H = np.zeros((6, 6))
    if len(pi) == 0:
        return 0.0, g, H, 0
    x4 = np.concatenate([x[pi], np.zeros((len(pi), 1), dtype=F)], axis=1)           # :399
    xj = (x4 @ j_ang.T).astype(F)                                                    # :403  [K,8]
    # point_gradient (4x6 float): identity block + 8 angular entries, :220-221, :405-412
    J = np.zeros((len(pi), 4, 6), dtype=F)
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1
    J[:, 1, 3] = xj[:, 0]; J[:, 2, 3] = xj[:, 1]
    J[:, 0, 4] = xj[:, 2]; J[:, 1, 4] = xj[:, 3]; J[:, 2, 4] = xj[:, 4]
    J[:, 0, 5] = xj[:, 5]; J[:, 1, 5] = xj[:, 6]; J[:, 2, 5] = xj[:, 7]
    # x_trans - mean in double, then to float (:253, :490)
    xt4 = np.zeros((len(pi), 4), dtype=F)
    xt4[:, :3] = (xt[pi].astype(np.float64) - grid.mean[vi]).astype(F)
    C = np.zeros((len(pi), 4, 4), dtype=F)
    C[:, :3, :3] = grid.icov[vi].astype(F)                                           # :491-492
    gd2 = F(d2)
    xC = np.einsum("ki,kij->kj", xt4, C).astype(F)                                   # x_trans4 * c_inv4
    q = np.einsum("ki,ki->k", xt4, xC).astype(F)
    e = np.exp(-gd2 * q * F(0.5)).astype(F)                                          # :497
    score_inc = (-d1 * e.astype(np.float64)).astype(F)                               # :499 (double product stored as float)
    e2 = (gd2 * e).astype(F)                                                         # :501
    ok = ~((e2 > 1) | (e2 < 0) | np.isnan(e2))                                       # :504-505
    e2 = (d1 * e2.astype(np.float64)).astype(F)                                      # :508
    CJ = np.einsum("kij,kjl->kil", C, J).astype(F)                                   # :510  [K,4,6]
    xCJ = np.einsum("ki,kil->kl", xt4, CJ).astype(F)                                 # :511  [K,6]
    score = float(score_inc[ok].astype(np.float64).sum())
    g = (e2[ok, None] * xCJ[ok]).astype(F).astype(np.float64).sum(axis=0)            # :513
    if compute_hessian:
        xh = (x4 @ h_ang.T).astype(F)                                                # :416  [K,16]
        # point_hessian: 4-vectors at (row block i, col j) for i,j in 3..5, :418-437
        z = np.zeros(len(pi), dtype=F)
        a = np.stack([z, xh[:, 0], xh[:, 1], z], axis=1)
        b = np.stack([z, xh[:, 2], xh[:, 3], z], axis=1)
        c = np.stack([z, xh[:, 4], xh[:, 5], z], axis=1)
        d = np.stack([xh[:, 6], xh[:, 7], xh[:, 8], z], axis=1)
        ee = np.stack([xh[:, 9], xh[:, 10], xh[:, 11], z], axis=1)
        f = np.stack([xh[:, 12], xh[:, 13], xh[:, 14], z], axis=1)
        PH = np.zeros((len(pi), 6, 6, 4), dtype=F)        # [k, i, j, :]
        PH[:, 3, 3] = a; PH[:, 4, 3] = b; PH[:, 5, 3] = c
        PH[:, 3, 4] = b; PH[:, 4, 4] = d; PH[:, 5, 4] = ee
        PH[:, 3, 5] = c; PH[:, 4, 5] = ee; PH[:, 5, 5] = f
        xCH = np.einsum("kc,kijc->kij", xC, PH).astype(F)                            # :523
        JCJ = np.einsum("kci,kcj->kij", J, CJ).astype(F)                             # :517  (j, i) indexed below
        term = (-gd2 * xCJ[:, :, None] * xCJ[:, None, :] + xCH + np.transpose(JCJ, (0, 2, 1))).astype(F)   # :527-529
        H = (e2[ok, None, None] * term[ok]).astype(F).astype(np.float64).sum(axis=0)
    return score, g, H, int(ok.sum())


# --------------------------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------------------------
def euler_xyz_from_matrix(R):
    """Eigen 3.3 MatrixBase::eulerAngles(0, 1, 2) (R = Rx(a) Ry(b) Rz(c); first angle in [0, pi])."""
    R = np.asarray(R, dtype=np.float64)
    i, j, k = 0, 1, 2                      # odd = 0 for (0, 1, 2)
    r0 = np.arctan2(R[j, k], R[k, k])
    c2 = np.hypot(R[i, i], R[i, j])
    if r0 > 0.0:                           # (!odd && res[0] > 0)
        r0 -= np.pi
        r1 = np.arctan2(-R[i, k], -c2)
    else:
        r1 = np.arctan2(-R[i, k], c2)
    s1, c1 = np.sin(r0), np.cos(r0)
    r2 = np.arctan2(s1 * R[k, i] - c1 * R[j, i], c1 * R[j, j] - s1 * R[k, j])
    return -np.array([r0, r1, r2])         # if (!odd) res = -res


def pose_to_matrix_f32(p):
    """Translation(p0..2) * Rx * Ry * Rz in float (ndt_omp_impl.hpp:146-149, 808-811)."""
    p = np.asarray(p, dtype=np.float64)
    a, b, c = (F(p[3]), F(p[4]), F(p[5]))
    ca, sa, cb, sb, cc, sc = (np.cos(a, dtype=F), np.sin(a, dtype=F), np.cos(b, dtype=F), np.sin(b, dtype=F),
                              np.cos(c, dtype=F), np.sin(c, dtype=F))
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]], dtype=F)
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]], dtype=F)
    Rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]], dtype=F)
    T = np.eye(4, dtype=F)
    T[:3, :3] = (Rx @ Ry @ Rz).astype(F)
    T[:3, 3] = p[:3].astype(F)
    return T


def transform_cloud_f32(pts_f32, T_f32):
    """pcl::transformPointCloud with a float 4x4."""
    x = np.asarray(pts_f32, dtype=F)[:, :3]
    T = np.asarray(T_f32, dtype=F)
    return (x @ T[:3, :3].T + T[:3, 3]).astype(F)


def svd_solve(H, b):
    """Eigen::JacobiSVD(H, FullU|FullV).solve(b): pseudo-inverse with Eigen's default threshold."""
    U, s, Vt = np.linalg.svd(H)
    thr = np.finfo(np.float64).eps * max(H.shape) * (s[0] if len(s) else 0.0)
    sinv = np.where(s > thr, 1.0 / np.where(s > thr, s, 1.0), 0.0)
    return Vt.T @ (sinv * (U.T @ b))


# --------------------------------------------------------------------------------------------
# More-Thuente pieces (ndt_omp_impl.hpp:633-753)
# --------------------------------------------------------------------------------------------
def _psi(a, f_a, f_0, g_0, mu): return f_a - f_0 - mu * g_0 * a
def _dpsi(g_a, g_0, mu): return g_a - mu * g_0


def _update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    if f_t > f_l:
        return a_l, f_l, g_l, a_t, f_t, g_t, False
    if g_t * (a_l - a_t) > 0:
        return a_t, f_t, g_t, a_u, f_u, g_u, False
    if g_t * (a_l - a_t) < 0:
        return a_t, f_t, g_t, a_l, f_l, g_l, False
    return a_l, f_l, g_l, a_u, f_u, g_u, True


def _trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    with np.errstate(all="ignore"):
        if f_t > f_l:
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t))
            return a_c if abs(a_c - a_l) < abs(a_q - a_l) else 0.5 * (a_q + a_c)
        if g_t * g_l < 0:
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
            return a_c if abs(a_c - a_t) >= abs(a_s - a_t) else a_s
        if abs(g_t) <= abs(g_l):
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
            nxt = a_c if abs(a_c - a_t) < abs(a_s - a_t) else a_s
            return min(a_t + 0.66 * (a_u - a_t), nxt) if a_t > a_l else max(a_t + 0.66 * (a_u - a_t), nxt)
        z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u
        w = np.sqrt(z * z - g_t * g_u)
        return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w)


def ndt_align(source_f32, target_f32, guess=None, resolution=1.0, step_size=0.1, outlier_ratio=0.55,
              trans_eps=0.1, max_iterations=35, grid: VoxelGrid | None = None, with_fitness=True, trace=None, real=None):
    """registrators/ndt.cc:38-64 -> pclomp computeTransformation (ndt_omp_impl.hpp:81-171).

    Returns dict(result 4x4 float64 (source->target), score = getFitnessScore() (mean squared 1-NN
    distance to the RAW target; lower is better), iterations, derivative_calls, trans_probability).
    """
    src = np.asarray(source_f32, dtype=F)[:, :3]
    tgt = np.asarray(target_f32, dtype=F)[:, :3]
    if grid is None:
        grid = VoxelGrid(tgt, resolution)
    d1, d2, _ = gauss_constants(resolution, outlier_ratio)
    G = np.eye(4, dtype=F) if guess is None else np.asarray(guess).astype(F)       # guess.cast<float>(), ndt.cc:58
    final = G.copy()
    trans = src.copy()
    if not np.array_equal(G, np.eye(4, dtype=F)):                                    # :95-101
        trans = transform_cloud_f32(src, G)
    p = np.zeros(6)
    p[:3] = final[:3, 3].astype(np.float64)                                          # :107-111
    p[3:] = euler_xyz_from_matrix(final[:3, :3]).astype(F).astype(np.float64)        # Vector3f eulerAngles
    calls = 0
    score, g, H, _ = compute_derivatives(grid, src, trans, p, d1, d2, True, real=real); calls += 1    # :119
    it = 0
    converged = False
    while not converged:                                                             # :121
        dp = svd_solve(H, -g)                                                        # :127-129
        dp_norm = np.linalg.norm(dp)
        if dp_norm == 0 or dp_norm != dp_norm:                                       # :134-139
            break
        step_dir = dp / dp_norm                                                      # :141
        # ---- computeStepLengthMT(p, step_dir, dp_norm, step_size, trans_eps / 2, ...)  :757-916
        step_init, step_max, step_min = dp_norm, step_size, trans_eps / 2
        phi_0 = -score
        d_phi_0 = -(g @ step_dir)
        a_t = 0.0
        skip = False
        if d_phi_0 >= 0:
            if d_phi_0 == 0:
                skip = True
            else:
                d_phi_0 *= -1
                step_dir = -step_dir
        if not skip:
            mu, nu = 1e-4, 0.9
            a_l = a_u = 0.0
            f_l = _psi(a_l, phi_0, phi_0, d_phi_0, mu); g_l = _dpsi(d_phi_0, d_phi_0, mu)
            f_u, g_u = f_l, g_l
            interval_converged = (step_max - step_min) > 0                           # :795 (sic)
            open_interval = True
            a_t = max(min(step_init, step_max), step_min)                            # :797-799
            x_t = p + step_dir * a_t
            final = pose_to_matrix_f32(x_t)                                          # :803-806
            trans = transform_cloud_f32(src, final)                                  # :809
            score, g, H, _ = compute_derivatives(grid, src, trans, x_t, d1, d2, True, real=real); calls += 1   # :813
            phi_t = -score; d_phi_t = -(g @ step_dir)
            psi_t = _psi(a_t, phi_t, phi_0, d_phi_0, mu); d_psi_t = _dpsi(d_phi_t, d_phi_0, mu)
            step_iterations = 0
            while (not interval_converged) and step_iterations < 10 and not (psi_t <= 0 and d_phi_t <= -nu * d_phi_0):
                if open_interval:
                    a_t = _trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                else:
                    a_t = _trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t)
                a_t = max(min(a_t, step_max), step_min)
                x_t = p + step_dir * a_t
                final = pose_to_matrix_f32(x_t)
                trans = transform_cloud_f32(src, final)
                score, g, _, _ = compute_derivatives(grid, src, trans, x_t, d1, d2, False, real=real); calls += 1
                phi_t = -score; d_phi_t = -(g @ step_dir)
                psi_t = _psi(a_t, phi_t, phi_0, d_phi_0, mu); d_psi_t = _dpsi(d_phi_t, d_phi_0, mu)
                if open_interval and (psi_t <= 0 and d_psi_t >= 0):
                    open_interval = False
                    f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0
                    f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0
                if open_interval:
                    a_l, f_l, g_l, a_u, f_u, g_u, interval_converged = _update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                else:
                    a_l, f_l, g_l, a_u, f_u, g_u, interval_converged = _update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t)
                step_iterations += 1
            if step_iterations:                                                      # :912-913
                _, _, H, _ = compute_derivatives(grid, src, trans, x_t, d1, d2, True, real=real); calls += 1
        dp_norm = a_t                                                                # :142
        dp = step_dir * dp_norm                                                      # :143
        p = p + dp                                                                   # :152
        if trace is not None:
            trace.append(dict(p=p.copy(), score=score, step=dp_norm))
        if it > max_iterations or (it and abs(dp_norm) < trans_eps):                 # :158-162
            converged = True
        it += 1                                                                      # :164
    result = final.astype(np.float64)                                                # getFinalTransformation().cast<double>()
    out = dict(result=result, iterations=it, derivative_calls=calls, trans_probability=score / len(src), p=p)
    if with_fitness:                                                                 # pcl::Registration::getFitnessScore, ndt.cc:60
        t = transform_cloud_f32(src, final)
        dist, _ = cKDTree(tgt.astype(np.float64)).query(t.astype(np.float64))
        out["score"] = float((dist.astype(F) ** 2).astype(np.float64).mean())
    return out
