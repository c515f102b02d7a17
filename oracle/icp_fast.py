"""numpy/scipy float64 restatement of `registrators::IcpFast` (TEST ORACLE).

PARITY UNPINNED (see oracle/__init__.py).  Every function cites the reference
lines it restates (paths relative to /root/reference).

Deliberate, documented deviations from the reference:
  * Nearest neighbour is EXACT (eps = 0); the reference asks libnabo for an
    approximate neighbour with eps = 3.16 (registrators/icp_fast.cc:174).
    libnabo is not available and its traversal order is not specified, so the
    oracle, the CPU baseline and the GPU path all use the exact rule
    (SURVEY.md §7 "hard parts").  `nn_eps` lets a test quantify the effect
    with scipy's (1+eps) contract.
  * `CalculateNormals`: the reference stores each leaf's result in column
    `indices[first]`, an artefact of std::nth_element's internal permutation;
    the oracle uses the smallest original index in the leaf.  Only the ORDER of
    the surviving target points can differ, never the set.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree
import scipy.linalg

K_NORMAL_KNN = 7          # builder/data/cloud_types.cc:38


# --------------------------------------------------------------------------
# EigenPointCloud helpers
# --------------------------------------------------------------------------
def eigen_cloud_from_inner(points_f32: np.ndarray) -> np.ndarray:
    """builder/data/cloud_types.cc:328-345 (FromPointCloud): f32 xyz -> f64 [N,3]."""
    return np.asarray(points_f32, dtype=np.float32)[:, :3].astype(np.float64)


def apply_transform(points: np.ndarray, T: np.ndarray) -> np.ndarray:
    """builder/data/cloud_types.cc:288-302 (ApplyTransform): p <- R p + t."""
    return points @ T[:3, :3].T + T[:3, 3]


def _arg_max_ref(v: np.ndarray) -> int:
    """builder/data/cloud_types.cc:41-56: custom ArgMax starting from (0, 0.)."""
    max_val, max_idx = 0.0, 0
    for i in range(v.shape[0]):
        if v[i] > max_val:
            max_val, max_idx = v[i], i
    return max_idx


def _inverse3_cofactor(M: np.ndarray) -> np.ndarray:
    """Eigen's fixed-size 3x3 inverse (cofactors times 1/det), used at cloud_types.cc:94."""
    c00 = M[1, 1] * M[2, 2] - M[1, 2] * M[2, 1]
    c01 = M[1, 2] * M[2, 0] - M[1, 0] * M[2, 2]
    c02 = M[1, 0] * M[2, 1] - M[1, 1] * M[2, 0]
    det = M[0, 0] * c00 + M[0, 1] * c01 + M[0, 2] * c02
    inv = np.array([
        [c00, M[0, 2] * M[2, 1] - M[0, 1] * M[2, 2], M[0, 1] * M[1, 2] - M[0, 2] * M[1, 1]],
        [c01, M[0, 0] * M[2, 2] - M[0, 2] * M[2, 0], M[0, 2] * M[1, 0] - M[0, 0] * M[1, 2]],
        [c02, M[0, 1] * M[2, 0] - M[0, 0] * M[2, 1], M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]]])
    return inv / det


def calculate_normals(points: np.ndarray):
    """builder/data/cloud_types.cc:347-368 + BuildNormals :105-144 + leaf :73-103.

    Returns (kept_points[M,3], kept_normals[M,3], leaf_sizes[M]) ordered by the
    smallest original index of each surviving leaf.
    """
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    indices = np.arange(n)
    leaves = []                     # (first, last)
    stack = [(0, n, pts.min(axis=0).copy(), pts.max(axis=0).copy())]
    while stack:
        first, last, lo, hi = stack.pop()
        count = last - first
        if count <= K_NORMAL_KNN:                                # :107-112
            leaves.append((first, last))
            continue
        cut_dim = _arg_max_ref(hi - lo)                          # :115
        right = count // 2                                       # :118
        left = count - right
        seg = indices[first:last]
        order = np.argpartition(pts[seg, cut_dim], left)         # :122-125 (nth_element at first+left)
        indices[first:last] = seg[order]
        cut_val = pts[indices[first + left], cut_dim]            # :128-129
        left_hi = hi.copy(); left_hi[cut_dim] = cut_val          # :132-133
        right_lo = lo.copy(); right_lo[cut_dim] = cut_val        # :135-136
        stack.append((first + left, last, right_lo, hi))
        stack.append((first, first + left, lo, left_hi))
    keep_idx, keep_pts, keep_nrm, keep_sz = [], [], [], []
    for first, last in leaves:
        ids = indices[first:last]
        d = pts[ids]                                             # :80-83
        if d.shape[0] == 0:
            continue
        M = d.T @ d                                              # :83
        b = d.sum(axis=0)                                        # :85
        mean = b / d.shape[0]
        nn = d - mean
        C = nn.T @ nn                                            # :89
        if np.linalg.matrix_rank(C) + 1 < 3:                     # :90-92
            continue
        with np.errstate(all="ignore"):
            normal = _inverse3_cofactor(M) @ b                    # :94  M.inverse() * b (Eigen 3x3 = cofactors)
            normal = normal / np.linalg.norm(normal)             # :102
        keep_idx.append(int(ids.min()))
        keep_pts.append(mean)
        keep_nrm.append(normal)
        keep_sz.append(d.shape[0])
    order = np.argsort(np.asarray(keep_idx), kind="stable")      # :358 sort(indices_to_keep)
    return (np.asarray(keep_pts)[order], np.asarray(keep_nrm)[order],
            np.asarray(keep_sz)[order])


# --------------------------------------------------------------------------
# IcpFast pieces
# --------------------------------------------------------------------------
def find_closests(tree: cKDTree, tgt: np.ndarray, pts: np.ndarray, nn_eps: float = 0.0):
    """registrators/icp_fast.cc:169-180: 1-NN ids + SQUARED distances."""
    _, ids = tree.query(pts, k=1, eps=nn_eps)
    diff = pts - tgt[ids]
    d2 = np.einsum("ij,ij->i", diff, diff)
    return ids.astype(np.int64), d2


def dists_quantile(d2: np.ndarray, quantile: float) -> float:
    """registrators/icp_fast.cc:65-90 (GetDistsQuantile) -- nth_element rank rule.

    `quantile` arrives as the float32 option widened to double
    (icp_fast.h:59, icp_fast.cc:496), so 0.7f = 0.699999988...: for n = 120000
    the index is 83999, not 84000.
    """
    assert 0.0 <= quantile <= 1.0
    vals = d2[d2 != np.inf]
    assert vals.size > 0
    if quantile == 1.0:
        return float(vals.max())
    k = int(vals.size * quantile)
    return float(np.partition(vals, k)[k])


def solve_possibly_underdetermined(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """registrators/icp_fast.cc:204-254."""
    # fullPivHouseholderQr().isInvertible(): all pivots > eps*size*max_pivot
    _, R, _ = scipy.linalg.qr(A, pivoting=True)
    piv = np.abs(np.diag(R))
    thresh = np.finfo(np.float64).eps * A.shape[0] * (piv.max() if piv.size else 0.0)
    rank = int((piv > thresh).sum())
    if rank == A.shape[0]:
        try:
            c, low = scipy.linalg.cho_factor(A)                   # :252 A.llt().solve(b)
            return scipy.linalg.cho_solve((c, low), b)
        except np.linalg.LinAlgError:
            pass
    # :215-249: smallest-norm solution of the rank-reduced system (== pinv)
    x, *_ = np.linalg.lstsq(A, b, rcond=thresh / (piv.max() if piv.max() > 0 else 1.0))
    return x


def angle_axis_matrix(v: np.ndarray) -> np.ndarray:
    """Eigen::AngleAxis(|v|, v/|v|).toRotationMatrix(), icp_fast.cc:309-310."""
    with np.errstate(all="ignore"):
        ang = np.linalg.norm(v)
        ax = v / ang
        c, s = np.cos(ang), np.sin(ang)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return c * np.eye(3) + s * K + (1 - c) * np.outer(ax, ax)


def compute_point_to_plane(reading: np.ndarray, ref_pts: np.ndarray, ref_nrm: np.ndarray):
    """registrators/icp_fast.cc:256-324 (weights are all 1 after compaction)."""
    cross = np.cross(reading, ref_nrm)                            # :267 / :182-202
    F = np.concatenate([cross, ref_nrm], axis=1)                  # :271-281   [K,6]
    A = F.T @ F                                                   # :292
    dot = np.einsum("ij,ij->i", reading - ref_pts, ref_nrm)       # :293-299
    b = -(F.T @ dot)                                              # :302
    x = solve_possibly_underdetermined(A, b)                      # :304
    T = np.eye(4)
    T[:3, :3] = angle_axis_matrix(x[:3])                          # :309-310
    T[:3, 3] = x[3:6]                                             # :312
    if np.isnan(T).any():                                         # :315-321
        T[:3, :3] = np.eye(3)
    return T, A, b


def _quat_from_matrix(R: np.ndarray) -> np.ndarray:
    """Eigen::Quaterniond(Matrix3d) (w, x, y, z)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[1 + i] = 0.25 * s
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


def _angular_distance(q1: np.ndarray, q2: np.ndarray) -> float:
    """Eigen::Quaternion::angularDistance: angle of q1 * conj(q2)."""
    w1, v1 = q1[0], q1[1:]
    w2, v2 = q2[0], -q2[1:]
    w = w1 * w2 - v1 @ v2
    v = w1 * v2 + w2 * v1 + np.cross(v1, v2)
    return 2.0 * np.arctan2(np.linalg.norm(v), abs(w))


def check_convergence(rotations, translations) -> bool:
    """registrators/icp_fast.cc:377-405."""
    k_smooth = 4
    if len(rotations) <= k_smooth:
        return False
    rd = td = 0.0
    for i in range(len(rotations) - 1, len(rotations) - 1 - k_smooth, -1):
        rd += abs(_angular_distance(rotations[i], rotations[i - 1]))
        td += abs(np.linalg.norm(translations[i] - translations[i - 1]))
    return (rd / k_smooth) < 1e-3 and (td / k_smooth) < 1e-2


def icp_fast_align(source: np.ndarray, target: np.ndarray, target_normals: np.ndarray,
                   guess: np.ndarray | None = None, max_iteration: int = 100,
                   dist_outlier_ratio: float = 0.7, early_exit: bool = True,
                   nn_eps: float = 0.0, trace: list | None = None):
    """registrators/icp_fast.cc:455-529 (IcpFast::Align).

    source [Ns,3], target [Nt,3], target_normals [Nt,3] float64.
    Returns (result 4x4 source->target, score, iterations).
    `early_exit=False` disables CheckConvergence so exactly `max_iteration`
    iterations run (throughput configuration).
    """
    S = np.asarray(source, dtype=np.float64)
    Q = np.asarray(target, dtype=np.float64)
    Nrm = np.asarray(target_normals, dtype=np.float64)
    if guess is None:
        guess = np.eye(4)
    rho = float(np.float32(dist_outlier_ratio))                   # icp_fast.h:59 (float option)
    mu = Q.sum(axis=0) / Q.shape[0]                               # :457-458
    T_mean = np.eye(4); T_mean[:3, 3] = mu                        # :459-460
    Qc = Q - mu                                                   # :462
    tree = cKDTree(Qc)                                            # :464-467
    T_mean_inv = np.eye(4); T_mean_inv[:3, 3] = -mu
    G = T_mean_inv @ guess                                        # :469
    P0 = apply_transform(S, G)                                    # :470
    T_iter = np.eye(4)                                            # :473
    rots = [np.array([1.0, 0, 0, 0])]                             # :478-479
    trans = [np.zeros(3)]
    it = 0
    score = float("nan")
    while True:
        P = apply_transform(P0, T_iter)                           # :486-491
        ids, d2 = find_closests(tree, Qc, P, nn_eps)              # :493
        limit = dists_quantile(d2, rho)                           # :496
        keep = (d2 <= limit) & (d2 != np.inf)                     # :497-498, :124-128
        reading = P[keep]                                         # :121-136
        ref_p = Qc[ids[keep]]                                     # :148-156
        ref_n = Nrm[ids[keep]]
        dT, A, b = compute_point_to_plane(reading, ref_p, ref_n)  # :506-510
        if trace is not None:
            trace.append(dict(ids=ids, d2=d2, limit=limit, keep=keep, A=A, b=b,
                              T_before=T_iter.copy(), dT=dT))
        T_iter = dT @ T_iter
        it += 1                                                   # :513
        rots.append(_quat_from_matrix(T_iter[:3, :3]))            # :514-515
        trans.append(T_iter[:3, 3].copy())
        if (early_exit and check_convergence(rots, trans)) or it >= max_iteration:   # :516-517
            score = float(np.exp(-np.sqrt(d2[keep]).sum() / keep.sum()))             # :518-521
            break
    result = T_mean @ T_iter @ G                                  # :527
    return result, score, it


def se3_error(Ta: np.ndarray, Tb: np.ndarray):
    """(rotation angle of Ra Rb^T [rad], |ta - tb| [m]) -- SURVEY.md §8(d) metric."""
    R = Ta[:3, :3] @ Tb[:3, :3].T
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    s = np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2.0
    return float(np.arctan2(s, c)), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
