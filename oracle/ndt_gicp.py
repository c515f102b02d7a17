"""numpy/scipy restatement of `registrators::NdtWithGicp` -- TEST ORACLE, PARITY UNPINNED
(see oracle/__init__.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product path never does.  Paths are relative to /root/reference/registrators.

Restates
  ndt_gicp.cc:28-53        options (voxel_resolution 0.2, use_ndt, using_voxel_filter), NDT: eps 0.01, step 0.1,
                           resolution 1.0, 35 iterations; GICP: rotation epsilon 1e-3, 35 iterations
  ndt_gicp.cc:55-112       Align: ApproximateVoxelGrid on both clouds -> pcl NDT -> (fitness <= 1) pcl GICP ->
                           score exp(-fitness), result = GICP final transformation; else result = guess, false
Every arithmetic step of that chain lives in PCL, which is NOT vendored (apt `libpcl-dev`; "tested in pcl-1.7 and
pcl-1.8", README.md:38-40).  Pinned here to **PCL 1.8.1** (Ubuntu 18.04, setup/Dockerfile.bionic.base:13) and
restated from its published sources:
  pcl/filters/impl/approximate_voxel_grid.hpp   applyFilter / flush (512-entry hash history, hash
                                                (ix*7171 + iy*3079 + iz*4231) & 511, float centroids)
  pcl/registration/impl/ndt.hpp                 same algorithm as the in-tree fork pclomp/ndt_omp_impl.hpp (restated
                                                in oracle/ndt.py) with the per-neighbour math in double
  pcl/registration/impl/gicp.hpp                the in-tree fork pclomp/gicp_omp_impl.hpp:59-131 (covariances),
                                                :133-186 (computeRDerivative), :189-247 (estimateRigidTransformationBFGS),
                                                :250-377 (functor f / df / fdf), :381-514 (computeTransformation),
                                                :516-527 (applyState) carries the same statements; lines cited below
                                                are the fork's
  pcl/registration/bfgs.h                       BFGS<Functor>: a port of GSL's vector_bfgs2 minimiser
                                                (Fletcher's line search, gsl multimin/linear_minimize.c)
  pcl::Registration::getFitnessScore            mean squared 1-NN distance of the transformed source to the target
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

from . import ndt as _ndt

F = np.float32
HIST = 512


# --------------------------------------------------------------------------------------------
# pcl::ApproximateVoxelGrid<PointXYZ>::applyFilter
# --------------------------------------------------------------------------------------------
def approximate_voxel_grid(points_f32, leaf=0.2):
    """The literal sequential loop.  Output = centroids in flush order (float32 [M,3])."""
    pts = np.asarray(points_f32, dtype=F)[:, :3]
    inv = F(1.0) / F(leaf)                                   # inverse_leaf_size_ = Array3f::Ones() / leaf_size_
    ijk = np.floor(pts * inv).astype(np.int64)               # float product, floor
    hsh = ((ijk[:, 0] * 7171 + ijk[:, 1] * 3079 + ijk[:, 2] * 4231) & (HIST - 1)).tolist()
    ijk_l = [tuple(r) for r in ijk.tolist()]
    cnt = [0] * HIST
    key = [None] * HIST
    acc = [np.zeros(3, dtype=F) for _ in range(HIST)]
    out = []
    for cp in range(len(pts)):
        h = hsh[cp]
        if cnt[h] and key[h] != ijk_l[cp]:
            out.append(acc[h] / F(cnt[h]))                   # flush: centroid /= float(count)
            cnt[h] = 0
            acc[h] = np.zeros(3, dtype=F)
        key[h] = ijk_l[cp]
        cnt[h] += 1
        acc[h] = acc[h] + pts[cp]                            # float accumulation in arrival order
    for h in range(HIST):
        if cnt[h]:
            out.append(acc[h] / F(cnt[h]))
    return np.asarray(out, dtype=F).reshape(-1, 3)


def approximate_voxel_grid_runs(points_f32, leaf=0.2):
    """The same filter restated without the serial dependence (the formulation the device uses): every hash
    bucket sees its own subsequence of points; maximal runs of equal voxel inside a bucket's subsequence are the
    flushed centroids; a run is flushed when the first point of the bucket's NEXT run arrives, the last run of
    each bucket at the end in bucket order.  Returns the same array as `approximate_voxel_grid`."""
    pts = np.asarray(points_f32, dtype=F)[:, :3]
    n = len(pts)
    inv = F(1.0) / F(leaf)
    ijk = np.floor(pts * inv).astype(np.int64)
    hsh = (ijk[:, 0] * 7171 + ijk[:, 1] * 3079 + ijk[:, 2] * 4231) & (HIST - 1)
    order = np.argsort(hsh, kind="stable")                   # by bucket, arrival order inside
    hs, vs = hsh[order], ijk[order]
    first = np.ones(n, dtype=bool)
    first[1:] = (hs[1:] != hs[:-1]) | np.any(vs[1:] != vs[:-1], axis=1)
    starts = np.flatnonzero(first)
    ends = np.append(starts[1:], n)
    # flush time: arrival index of the next run's first point in the same bucket, else n + bucket
    nxt_same = np.zeros(len(starts), dtype=bool)
    nxt_same[:-1] = hs[starts[1:]] == hs[starts[:-1]]
    when = np.where(nxt_same, order[np.minimum(ends, n - 1)], n + hs[starts])
    cent = np.zeros((len(starts), 3), dtype=F)
    for r, (a, b) in enumerate(zip(starts, ends)):
        s = np.zeros(3, dtype=F)
        for k in order[a:b]:
            s = s + pts[k]
        cent[r] = s / F(b - a)
    return cent[np.argsort(when, kind="stable")]


# --------------------------------------------------------------------------------------------
# GICP
# --------------------------------------------------------------------------------------------
def gicp_covariances(cloud_f32, k=20, gicp_epsilon=1e-3, return_nn=False):
    """gicp_omp_impl.hpp:59-131.  Single-pass covariance of the k nearest neighbours (the point itself
    included) with the products pt.x * pt.y taken in FLOAT before the double accumulation (:95-103), then
    U diag(1, 1, eps) U^T from the SVD (:118-129)."""
    pf = np.asarray(cloud_f32, dtype=F)[:, :3]
    pd = pf.astype(np.float64)
    _, nn = cKDTree(pd).query(pd, k=k)
    P = pf[nn]                                               # [N, k, 3] float
    mean = P.astype(np.float64).sum(axis=1) / k
    prod = (P[:, :, :, None] * P[:, :, None, :])             # float products
    cov = prod.astype(np.float64).sum(axis=1) / k - mean[:, :, None] * mean[:, None, :]
    cov = 0.5 * (cov + np.transpose(cov, (0, 2, 1)))
    w, U = np.linalg.eigh(cov)
    # JacobiSVD orders by singular value = |eigenvalue| (the float products can push the single-pass covariance
    # slightly indefinite): the direction that gets gicp_epsilon is the one with the smallest |w|
    col = np.argmin(np.abs(w), axis=1)
    u3 = np.take_along_axis(U, col[:, None, None], axis=2)[:, :, 0]
    out = np.eye(3)[None] - (1.0 - gicp_epsilon) * u3[:, :, None] * u3[:, None, :]
    return (out, nn) if return_nn else out


def rotation_zyx_f32(x):
    """applyState (:516-527): AngleAxisf(x5, Z) * AngleAxisf(x4, Y) * AngleAxisf(x3, X) as a float matrix."""
    cx, sx = np.cos(F(x[3])), np.sin(F(x[3]))
    cy, sy = np.cos(F(x[4])), np.sin(F(x[4]))
    cz, sz = np.cos(F(x[5])), np.sin(F(x[5]))
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=F)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=F)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=F)
    return (Rz @ Ry @ Rx).astype(F)


def apply_state_f32(T_f32, x):
    """t.topLeft = R(x) * t.topLeft ; t.col(3) += (x0, x1, x2, 0)   (:516-527)"""
    T = np.array(T_f32, dtype=F)
    T[:3, :3] = (rotation_zyx_f32(x) @ T[:3, :3]).astype(F)
    T[:3, 3] = (T[:3, 3] + np.asarray(x[:3]).astype(F)).astype(F)
    return T


def r_derivative(x, R):
    """computeRDerivative (:133-186): g[3:6] = <dR/dphi, R>, <dR/dtheta, R>, <dR/dpsi, R> with
    matricesInnerProd(A, B) = sum_ij A(j, i) B(i, j)  (gicp_omp.h:318-327)."""
    phi, theta, psi = x[3], x[4], x[5]
    cphi, sphi = np.cos(phi), np.sin(phi)
    cth, sth = np.cos(theta), np.sin(theta)
    cpsi, spsi = np.cos(psi), np.sin(psi)
    dphi = np.array([[0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth],
                     [0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth],
                     [0, cphi * cth, -cth * sphi]])
    dth = np.array([[-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth],
                    [-spsi * sth, cth * sphi * spsi, cphi * cth * spsi],
                    [-cth, -sphi * sth, -cphi * sth]])
    dpsi = np.array([[-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth],
                     [cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth],
                     [0, 0, 0]])
    ip = lambda A, B: float(np.sum(A.T * B))
    return np.array([ip(dphi, R), ip(dth, R), ip(dpsi, R)])


class GicpFunctor:
    """OptimizationFunctorWithIndices (:250-377) over one fixed correspondence set."""

    def __init__(self, base_T_f32, src_f32, tgt_f32, maha, transform_mode="blas"):
        # transform_mode: how the float 4x4 * point product of :263-268 is rounded -- "blas" (numpy's sgemm), "nofma" (every
        # multiply and add rounded to float, what -ffp-contract=off compiles to) or "fma" (one rounding per multiply-add, what
        # -ffp-contract=fast / a GPU compiles to).  All three are legal compilations of the same PCL statement; they differ
        # by one float ulp in some coordinates, which is what tests/test_oracle_ndt_gicp.py uses to measure GICP's own
        # repeatability.
        self.transform_mode = transform_mode
        self.base = np.asarray(base_T_f32, dtype=F)
        self.src4 = np.concatenate([src_f32, np.ones((len(src_f32), 1), dtype=F)], axis=1).astype(F)
        self.tgt = np.asarray(tgt_f32, dtype=F)
        self.maha = maha
        self.m = len(src_f32)
        self.pbase = (self.src4 @ self.base.T).astype(F)[:, :3].astype(np.float64)   # base_transformation_ * p_src
        self.evals = 0
        self.points = 0        # distinct states evaluated: pcl's BFGS asks for f at a trial state and then, maybe, for df at the same one

    def _transform(self, T):
        if self.transform_mode == "blas":
            return (self.src4 @ T.T).astype(F)[:, :3]
        x, y, z = self.src4[:, 0], self.src4[:, 1], self.src4[:, 2]
        out = np.empty((self.m, 3), dtype=F)
        for r in range(3):
            a, b, c, d = T[r, 0], T[r, 1], T[r, 2], T[r, 3]
            if self.transform_mode == "nofma":
                out[:, r] = F(F(F(a * x) + F(b * y)) + F(c * z)) + d                       # every operation rounded to float
            else:                                                                         # fma(c, z, fma(b, y, fma(a, x, d)))
                t = (a.astype(np.float64) * x.astype(np.float64) + np.float64(d)).astype(F)    # exact product, one rounding
                t = (b.astype(np.float64) * y.astype(np.float64) + t.astype(np.float64)).astype(F)
                out[:, r] = (c.astype(np.float64) * z.astype(np.float64) + t.astype(np.float64)).astype(F)
        return out

    def _res_temp(self, x):
        T = apply_state_f32(self.base, x)
        pp = self._transform(T)
        res = (pp - self.tgt).astype(F).astype(np.float64)                           # float differences (:268)
        temp = np.einsum("kij,kj->ki", self.maha, res)
        return res, temp

    def f(self, x):
        self.evals += 1
        self.points += 1
        res, temp = self._res_temp(x)
        return float(np.einsum("ki,ki->", res, temp)) / self.m

    def df(self, x):
        self.evals += 1
        res, temp = self._res_temp(x)
        g = np.zeros(6)
        g[:3] = temp.sum(axis=0) * (2.0 / self.m)
        R = (self.pbase[:, :, None] * temp[:, None, :]).sum(axis=0) * (2.0 / self.m)
        g[3:] = r_derivative(x, R)
        return g

    def fdf(self, x):
        self.evals += 1
        self.points += 1
        res, temp = self._res_temp(x)
        f = float(np.einsum("ki,ki->", res, temp)) / self.m
        g = np.zeros(6)
        g[:3] = temp.sum(axis=0) * (2.0 / self.m)
        R = (self.pbase[:, :, None] * temp[:, None, :]).sum(axis=0) * (2.0 / self.m)
        g[3:] = r_derivative(x, R)
        return f, g


# ---- pcl/registration/bfgs.h (GSL vector_bfgs2) ---------------------------------------------
SUCCESS, NO_PROGRESS, RUNNING = 0, 1, -1
DBL_EPS = np.finfo(np.float64).eps


def _solve_quadratic(a, b, c):
    """gsl_poly_solve_quadratic: real roots of a x^2 + b x + c in ascending order."""
    if a == 0:
        if b == 0:
            return []
        return [-c / b]
    disc = b * b - 4 * a * c
    if disc > 0:
        if b == 0:
            r = np.sqrt(-c / a)
            return [-r, r]
        sgnb = 1.0 if b > 0 else -1.0
        temp = -0.5 * (b + sgnb * np.sqrt(disc))
        r1, r2 = temp / a, c / temp
        return [min(r1, r2), max(r1, r2)]
    if disc == 0:
        return [-0.5 * b / a, -0.5 * b / a]
    return []


def _interp_quad(f0, fp0, f1, zl, zh):
    fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0))
    fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0))
    c = 2 * (f1 - f0 - fp0)
    zmin, fmin = zl, fl
    if fh < fmin:
        zmin, fmin = zh, fh
    if c > 0:
        z = -fp0 / c
        if zl < z < zh:
            f = f0 + z * (fp0 + z * (f1 - f0 - fp0))
            if f < fmin:
                zmin, fmin = z, f
    return zmin


def _interp_cubic(f0, fp0, f1, fp1, zl, zh):
    eta = 3 * (f1 - f0) - 2 * fp0 - fp1
    xi = fp0 + fp1 - 2 * (f1 - f0)
    c0, c1, c2, c3 = f0, fp0, eta, xi
    cubic = lambda z: c0 + z * (c1 + z * (c2 + z * c3))
    zmin, fmin = zl, cubic(zl)
    for z, inside in [(zh, True)] + [(r, zl < r < zh) for r in _solve_quadratic(3 * c3, 2 * c2, c1)]:
        if inside:
            y = cubic(z)
            if y < fmin:
                zmin, fmin = z, y
    return zmin


def _interpolate(a, fa, fpa, b, fb, fpb, xmin, xmax, order):
    zmin = (xmin - a) / (b - a)
    zmax = (xmax - a) / (b - a)
    if zmin > zmax:
        zmin, zmax = zmax, zmin
    if order > 2 and np.isfinite(fpb):
        z = _interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax)
    else:
        z = _interp_quad(fa, fpa * (b - a), fb, zmin, zmax)
    return a + z * (b - a)


class Bfgs:
    def __init__(self, functor, sigma=0.01, rho=0.01, tau1=9.0, tau2=0.05, tau3=0.5, order=3, step_size=1.0):
        self.fn = functor
        self.sigma, self.rho, self.tau1, self.tau2, self.tau3, self.order, self.step_size = sigma, rho, tau1, tau2, tau3, order, step_size

    # -- the line-function wrapper with its one-entry caches
    def _move(self, alpha):
        if alpha != self.x_key:
            self.x_alpha = self.x0 + alpha * self.p
            self.x_key = alpha

    def _F(self, alpha):
        if alpha == self.f_key:
            return self.f_alpha
        self._move(alpha)
        self.f_alpha = self.fn.f(self.x_alpha)
        self.f_key = alpha
        return self.f_alpha

    def _DF(self, alpha):
        if alpha == self.df_key:
            return self.df_alpha
        self._move(alpha)
        if alpha != self.g_key:
            self.g_alpha = self.fn.df(self.x_alpha)
            self.g_key = alpha
        self.df_alpha = float(self.g_alpha @ self.p)
        self.df_key = alpha
        return self.df_alpha

    def _FDF(self, alpha):
        if alpha == self.f_key and alpha == self.df_key:
            return self.f_alpha, self.df_alpha
        if alpha == self.f_key or alpha == self.df_key:
            return self._F(alpha), self._DF(alpha)
        self._move(alpha)
        self.f_alpha, self.g_alpha = self.fn.fdf(self.x_alpha)
        self.f_key = self.g_key = alpha
        self.df_alpha = float(self.g_alpha @ self.p)
        self.df_key = alpha
        return self.f_alpha, self.df_alpha

    def init(self, x):
        self.delta_f = 0.0
        self.f, self.gradient = self.fn.fdf(x)
        self.x0 = x.copy()
        self.g0 = self.gradient.copy()
        self.g0norm = float(np.linalg.norm(self.g0))
        self.p = -self.gradient / self.g0norm
        self.pnorm = float(np.linalg.norm(self.p))
        self.fp0 = -self.g0norm
        self.x_alpha, self.x_key = self.x0.copy(), 0.0
        self.f_alpha, self.f_key = self.f, 0.0
        self.g_alpha, self.g_key = self.g0.copy(), 0.0
        self.df_alpha, self.df_key = float(self.g_alpha @ self.p), 0.0

    def _line_search(self, alpha1):
        rho, sigma, tau1, tau2, tau3, order = self.rho, self.sigma, self.tau1, self.tau2, self.tau3, self.order
        f0, fp0 = self._FDF(0.0)
        falpha_prev, fpalpha_prev = f0, fp0
        alpha, alpha_prev = alpha1, 0.0
        a, b, fa, fb, fpa, fpb = 0.0, alpha, f0, 0.0, fp0, 0.0
        i = 0
        dbg = getattr(self, "debug", False)
        if dbg:
            print("[oracle]     ls f0=%.15g fp0=%.9g alpha1=%.9g" % (f0, fp0, alpha1))
        while i < 100:                                           # bracketing
            i += 1
            falpha = self._F(alpha)
            if dbg:
                print("[oracle]     br alpha=%.12g f=%.15g" % (alpha, falpha))
            if falpha > f0 + alpha * rho * fp0 or falpha >= falpha_prev:
                a, fa, fpa = alpha_prev, falpha_prev, fpalpha_prev
                b, fb, fpb = alpha, falpha, np.nan
                break
            fpalpha = self._DF(alpha)
            if abs(fpalpha) <= -sigma * fp0:
                return SUCCESS, alpha
            if fpalpha >= 0:
                a, fa, fpa = alpha, falpha, fpalpha
                b, fb, fpb = alpha_prev, falpha_prev, fpalpha_prev
                break
            delta = alpha - alpha_prev
            alpha_next = _interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha,
                                      alpha + delta, alpha + tau1 * delta, order)
            alpha_prev, falpha_prev, fpalpha_prev = alpha, falpha, fpalpha
            alpha = alpha_next
        else:
            i += 1
        while i < 100:                                           # sectioning (the counter is shared, as in GSL)
            i += 1
            delta = b - a
            alpha = _interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order)
            falpha = self._F(alpha)
            if dbg and i < 12:
                print("[oracle]     sec a=%.9g b=%.9g alpha=%.12g f=%.15g fa=%.15g fpa=%.6g" % (a, b, alpha, falpha, fa, fpa))
            if (a - alpha) * fpa <= DBL_EPS:
                return NO_PROGRESS, alpha
            if falpha > f0 + rho * alpha * fp0 or falpha >= fa:
                b, fb, fpb = alpha, falpha, np.nan
            else:
                fpalpha = self._DF(alpha)
                if abs(fpalpha) <= -sigma * fp0:
                    return SUCCESS, alpha
                if ((b - a) >= 0 and fpalpha >= 0) or ((b - a) <= 0 and fpalpha <= 0):
                    b, fb, fpb = a, fa, fpa
                a, fa, fpa = alpha, falpha, fpalpha
        return SUCCESS, alpha

    def one_step(self, x):
        f0 = self.f
        if self.pnorm == 0.0 or self.g0norm == 0.0 or self.fp0 == 0:
            return NO_PROGRESS, x
        if self.delta_f < 0:
            dl = max(-self.delta_f, 10 * DBL_EPS * abs(f0))
            alpha1 = min(1.0, 2.0 * dl / (-self.fp0))
        else:
            alpha1 = abs(self.step_size)
        status, alpha = self._line_search(alpha1)
        if status != SUCCESS:
            return status, x
        self.f, _ = self._FDF(alpha)                              # updatePosition
        x = self.x_alpha.copy()
        self.gradient = self.g_alpha.copy()
        self.delta_f = self.f - f0
        dx0 = x - self.x0
        dg0 = self.gradient - self.g0
        dxg, dgg, dxdg = dx0 @ self.gradient, dg0 @ self.gradient, dx0 @ dg0
        dgnorm = np.linalg.norm(dg0)
        if dxdg != 0:
            B = dxg / dxdg
            A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg
        else:
            A = B = 0.0
        p = self.gradient - A * dx0 - B * dg0
        self.g0 = self.gradient.copy()
        self.x0 = x.copy()
        self.g0norm = float(np.linalg.norm(self.g0))
        self.pnorm = float(np.linalg.norm(p))
        direction = -1.0 if (p @ self.gradient) > 0 else 1.0
        self.p = p * (direction / self.pnorm)
        self.pnorm = float(np.linalg.norm(self.p))
        self.fp0 = float(self.p @ self.g0)
        # changeDirection
        self.x_alpha, self.x_key = self.x0.copy(), 0.0
        self.f_key = 0.0
        self.g_alpha, self.g_key = self.g0.copy(), 0.0
        self.df_alpha, self.df_key = float(self.g_alpha @ self.p), 0.0
        return SUCCESS, x

    def test_gradient(self, eps):
        return SUCCESS if self.g0norm < eps else RUNNING


def estimate_rigid_transformation_bfgs(functor, T_f32, max_inner_iterations=20, gradient_tol=1e-2, debug=False):
    """:189-247.  x from the current transformation_, <= 20 BFGS steps, transformation_ = applyState(I, x)."""
    T = np.asarray(T_f32, dtype=F)
    x = np.zeros(6)
    x[:3] = T[:3, 3]
    x[3] = np.arctan2(T[2, 1], T[2, 2])                           # float atan2 arguments promoted to double
    x[4] = np.arcsin(-np.float64(T[2, 0]))
    x[5] = np.arctan2(T[1, 0], T[0, 0])
    bfgs = Bfgs(functor)
    bfgs.debug = debug
    bfgs.init(x)
    inner = 0
    if debug:
        print("[oracle] f0=%.12g g0=%s" % (bfgs.f, np.array2string(bfgs.g0, precision=6)))
    while True:
        inner += 1
        result, x = bfgs.one_step(x)
        if debug:
            print("[oracle]   inner %d status %d f=%.12g |g|=%.6g x=%s evals=%d" % (inner, result, bfgs.f, bfgs.g0norm, np.array2string(x, precision=8), functor.evals))
        if result:
            break
        result = bfgs.test_gradient(gradient_tol)
        if not (result == RUNNING and inner < max_inner_iterations):
            break
    return apply_state_f32(np.eye(4, dtype=F), x), x, inner


def gicp_align(src_f32, tgt_f32, guess_f32, max_iterations=35, rotation_epsilon=1e-3, transformation_epsilon=5e-4,
               corr_dist_threshold=5.0, k=20, gicp_epsilon=1e-3, max_inner_iterations=20, trace=None, debug=False,
               transform_mode="blas"):
    """computeTransformation (:381-514) + getFitnessScore.  Returns dict(result f32 4x4, score, iterations)."""
    src = np.asarray(src_f32, dtype=F)[:, :3]
    tgt = np.asarray(tgt_f32, dtype=F)[:, :3]
    guess = np.asarray(guess_f32, dtype=F)
    C_t = gicp_covariances(tgt, k, gicp_epsilon)
    C_s = gicp_covariances(src, k, gicp_epsilon)
    tree = cKDTree(tgt.astype(np.float64))
    src4 = np.concatenate([src, np.ones((len(src), 1), dtype=F)], axis=1)
    T = np.eye(4, dtype=F)                                        # transformation_
    prev = T.copy()
    it = 0
    while True:
        TR = T.astype(np.float64) @ guess.astype(np.float64)     # :425-429
        R = TR[:3, :3]
        q = (src4 @ guess.T).astype(F)                            # :439-440 (two float products)
        q = (q @ T.T).astype(F)[:, :3]
        d, j = tree.query(q.astype(np.float64))
        d2 = (d.astype(F) ** 2)                                   # float squared distances from the kd-tree
        keep = np.flatnonzero(d2 < F(corr_dist_threshold * corr_dist_threshold))      # :449
        Mh = np.linalg.inv(R[None] @ C_s[keep] @ R.T[None] + C_t[j[keep]])            # :451-459
        prev = T.copy()
        fn = GicpFunctor(guess, src[keep], tgt[j[keep]], Mh, transform_mode=transform_mode)
        if len(keep) < 4:                                         # NotEnoughPointsException -> break (:494-498)
            break
        T, x, inner = estimate_rigid_transformation_bfgs(fn, T, max_inner_iterations, debug=debug)
        ratio = np.full((4, 4), 1.0 / transformation_epsilon)
        ratio[:3, :3] = 1.0 / rotation_epsilon
        delta = float(np.max(ratio * np.abs(prev.astype(np.float64) - T.astype(np.float64))))   # :475-491
        it += 1
        if trace is not None:
            trace.append(dict(x=x.copy(), inner=inner, delta=delta, n_corr=len(keep), evals=fn.evals, points=fn.points))
        if it >= max_iterations or delta < 1:
            prev = T.copy()
            break
    final = np.eye(4, dtype=F)
    final[:3, :3] = (prev[:3, :3] @ guess[:3, :3]).astype(F)      # :506-509
    final[:3, 3] = (prev[:3, 3] + guess[:3, 3]).astype(F)
    t = (src4 @ final.T).astype(F)[:, :3]
    dist, _ = tree.query(t.astype(np.float64))
    score = float((dist.astype(F) ** 2).astype(np.float64).mean())
    return dict(result=final, score=score, iterations=it)


# --------------------------------------------------------------------------------------------
# NdtWithGicp::Align
# --------------------------------------------------------------------------------------------
def ndt_gicp_align(source_f32, target_f32, guess=None, voxel_resolution=0.2, using_voxel_filter=True, use_ndt=True,
                   downsampled=None):
    """ndt_gicp.cc:55-112.  Returns dict(ok, result 4x4 float64, score = exp(-fitness), ndt=..., gicp=...,
    n_source, n_target).  `downsampled` = (src, tgt) skips the filter (tests hand in the device's output when
    they check the later stages in isolation)."""
    G = np.eye(4) if guess is None else np.asarray(guess, dtype=np.float64)
    if downsampled is not None:
        ds, dt = downsampled
    elif using_voxel_filter:
        ds = approximate_voxel_grid(source_f32, voxel_resolution)
        dt = approximate_voxel_grid(target_f32, voxel_resolution)
    else:
        ds = np.asarray(source_f32, dtype=F)[:, :3]
        dt = np.asarray(target_f32, dtype=F)[:, :3]
    ndt_guess = G.astype(F)
    ndt_score = 0.9
    ndt_out = None
    if use_ndt:                                                   # :83-90
        ndt_out = _ndt.ndt_align(ds, dt, guess=G, resolution=1.0, step_size=0.1, trans_eps=0.01, max_iterations=35,
                                 real=np.float64)
        ndt_score = ndt_out["score"]
        ndt_guess = ndt_out["result"].astype(F)
    out = dict(n_source=len(ds), n_target=len(dt), ndt=ndt_out, gicp=None)
    if ndt_score <= 1.0:                                          # :94
        g = gicp_align(ds, dt, ndt_guess)
        out.update(ok=True, result=g["result"].astype(np.float64), score=float(np.exp(-g["score"])), gicp=g)
    else:
        out.update(ok=False, result=G.copy(), score=float(np.exp(-10.0)))
    return out
