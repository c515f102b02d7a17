"""Restatement of registrators::IcpUsingPointMatcher (TEST ORACLE, PARITY UNPINNED).

/root/reference/registrators/icp_pointmatcher.cc:104-149 (Align) and :166-247 (the libpointmatcher
1.3.1 chain).  libpointmatcher is not available here; the chain is restated from its configuration:
it is the same algorithm IcpFast hand-writes (README.md:170), so the pieces of oracle/icp_fast.py
are reused.  Deviations: exact NN instead of eps = 3.16, float64 instead of float, and the
RandomSampling mask is an input (the reference draws it with an unseeded std::rand()).
"""
import numpy as np
from scipy.spatial import cKDTree

from . import icp_fast as o


def align(reading_f32, reference_f32, guess, keep_mask, normals_fn=None, use_c=False, nn_eps=None):
    """Returns (accepted, result 4x4, score, iterations).  use_c: run the ICP loop through the C restatement
    (oracle/csrc/smref_icp.c, same algorithm, seconds instead of minutes on submap-sized clouds)."""
    rd = np.asarray(reading_f32, dtype=np.float32)
    rf = np.asarray(reference_f32, dtype=np.float32)
    rd = rd[~np.isnan(rd[:, :3]).any(axis=1)][:, :3].astype(np.float64)      # :57-66
    rf = rf[~np.isnan(rf[:, :3]).any(axis=1)][:, :3].astype(np.float64)
    # reference filter SamplingSurfaceNormal(knn 7, samplingMethod 1)  == CalculateNormals         :178-184
    q, n, _ = (normals_fn or o.calculate_normals)(rf)
    ok = np.isfinite(n).all(axis=1)
    q, n = q[ok], n[ok]
    # compute(): Counter(150) + Differential checkers, TrimmedDist 0.7, PointToPlane               :187-224
    if use_c or nn_eps is not None:
        # nn_eps: the KDTreeMatcher as the reference configures it (knn 1, epsilon 3.16, :186-191) = libnabo's approximate search
        from . import cref
        r = cref.icp_fast_align(rd[keep_mask], q, n, guess=guess, max_iteration=150, dist_outlier_ratio=0.7, nthreads=cref.usable_cores(),
                                nn_eps=nn_eps)
        result, it = r["result"], r["iterations"]
    else:
        result, _, it = o.icp_fast_align(rd[keep_mask], q, n, guess=guess, max_iteration=150, dist_outlier_ratio=0.7)
    # final score: transformed FULL reading vs RAW reference, trimmed 0.7, mean distance           :112-143
    P = o.apply_transform(rd, result)
    if nn_eps is not None:
        from . import cref
        _, d2, _ = cref.nn_nabo(rf, P, nn_eps)                 # matcher->findClosests with the same epsilon (:115-118)
    else:
        d, _ = cKDTree(rf).query(P)
        d2 = d * d
    limit = o.dists_quantile(d2, float(np.float32(0.7)))
    kept = d2 <= limit
    score = float(np.exp(-np.sqrt(d2[kept]).sum() / kept.sum()))
    return score >= 0.6, result, score, it
