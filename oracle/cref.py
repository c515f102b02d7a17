"""ctypes binding of oracle/_build/libsmref.so (C restatement; TEST ORACLE and
timed CPU baseline only -- see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess
import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")      # read by libgomp when the library loads: idle threads sleep
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "_build", "libsmref.so")
    src = os.path.join(_HERE, "csrc", "smref_icp.c")
    stale = os.path.exists(path) and any(os.path.getmtime(os.path.join(_HERE, "csrc", f)) > os.path.getmtime(path)
                                          for f in os.listdir(os.path.join(_HERE, "csrc")))
    if force or stale or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        lp = ctypes.POINTER(ctypes.c_long)
        _LIB.smref_icp_align_ex.argtypes = [dp, ctypes.c_int, dp, dp, ctypes.c_int, dp, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, dp, ip, dp, ip, dp, lp]
        _LIB.smref_icp_align_ex.restype = ctypes.c_int
        _LIB.smref_nn_nabo.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, ctypes.c_double, ip, dp, lp]
        _LIB.smref_nn_nabo.restype = ctypes.c_int
        _LIB.smref_nn.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, ip, dp]
        _LIB.smref_nn.restype = ctypes.c_int
        _LIB.smref_calculate_normals.argtypes = [dp, ctypes.c_int, dp, dp, ip]
        _LIB.smref_calculate_normals.restype = ctypes.c_int
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can see
    256 CPUs and own 8 -- 256 spinning OpenMP threads are then far slower than one) and by the physical core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        cores = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if cores:
            n = min(n, len(cores))
    except Exception:
        pass
    return max(1, n)


def icp_fast_align(source, target, target_normals, guess=None, max_iteration=100,
                   dist_outlier_ratio=0.7, early_exit=True, nthreads=1, want_matches=False, nn_eps=None):
    """C restatement of IcpFast::Align.  Returns dict(result, score, iterations, block_times[, ids, d2]).
    nn_eps None: exact 1-NN (smallest-id tie rule; what the GPU path computes); nn_eps >= 0: libnabo's
    KDTREE_LINEAR_HEAP tree and eps-approximate search restated (3.16 = the reference's call, icp_fast.cc:174)."""
    src, psrc = _d(source)
    tgt, ptgt = _d(target)
    nrm, pnrm = _d(target_normals)
    g, pg = _d(np.eye(4) if guess is None else guess)
    res = np.zeros((4, 4))
    score = ctypes.c_double()
    iters = ctypes.c_int()
    bt = np.zeros(6)
    leaves = ctypes.c_long()
    ids = np.zeros(src.shape[0], dtype=np.int32) if want_matches else None
    d2 = np.zeros(src.shape[0]) if want_matches else None
    rc = lib().smref_icp_align_ex(
        psrc, src.shape[0], ptgt, pnrm, tgt.shape[0], pg, int(max_iteration),
        ctypes.c_float(dist_outlier_ratio), int(early_exit), int(nthreads), -1.0 if nn_eps is None else float(nn_eps),
        res.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(score), ctypes.byref(iters),
        bt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if want_matches else None,
        d2.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if want_matches else None, ctypes.byref(leaves))
    if rc != 0:
        raise RuntimeError(f"smref_icp_align failed: {rc}")
    out = dict(result=res, score=score.value, iterations=iters.value, leaves_visited=leaves.value,
               block_times=dict(FindClosests=bt[0], ErrorElements=bt[1], ComputePointToPlane=bt[2], BuildKdTree=bt[3],
                                ApplyTransform=bt[4], Align=bt[5]))
    if want_matches:
        out["ids"], out["d2"] = ids, d2
    return out


def nn(target, query):
    tgt, ptgt = _d(target)
    q, pq = _d(query)
    ids = np.zeros(q.shape[0], dtype=np.int32)
    d2 = np.zeros(q.shape[0])
    lib().smref_nn(ptgt, tgt.shape[0], pq, q.shape[0], ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                   d2.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return ids, d2


def nn_nabo(target, query, eps=3.16):
    """libnabo KDTREE_LINEAR_HEAP knn(k = 1, epsilon = eps) restated; returns (ids, squared distances, leaves visited)."""
    tgt, ptgt = _d(target)
    q, pq = _d(query)
    ids = np.zeros(q.shape[0], dtype=np.int32)
    d2 = np.zeros(q.shape[0])
    leaves = ctypes.c_long()
    lib().smref_nn_nabo(ptgt, tgt.shape[0], pq, q.shape[0], float(eps), ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                        d2.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(leaves))
    return ids, d2, leaves.value


def calculate_normals(points):
    pts, pp = _d(points)
    n = pts.shape[0]
    op = np.zeros((n, 3)); on = np.zeros((n, 3)); sz = np.zeros(n, dtype=np.int32)
    m = lib().smref_calculate_normals(pp, n, op.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      on.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return op[:m].copy(), on[:m].copy(), sz[:m].copy()


# ------------------------------------------------------------------------------------------------------------
# registrators::Ndt (oracle/csrc/smref_ndt.c)
# ------------------------------------------------------------------------------------------------------------
def _ndt_lib():
    L = lib()
    if not getattr(L, "_ndt_ready", False):
        fp = ctypes.POINTER(ctypes.c_float); dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
        lp = ctypes.POINTER(ctypes.c_long)
        L.smref_ndt_grid_build.argtypes = [fp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_double]
        L.smref_ndt_grid_build.restype = ctypes.c_void_p
        L.smref_ndt_grid_free.argtypes = [ctypes.c_void_p]
        L.smref_ndt_grid_size.argtypes = [ctypes.c_void_p]
        L.smref_ndt_grid_get.argtypes = [ctypes.c_void_p, lp, dp, dp, fp, ip]
        L.smref_ndt_compute_derivatives.argtypes = [ctypes.c_void_p, fp, fp, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int,
                                                    ctypes.c_int, dp, dp, dp]
        L.smref_ndt_compute_derivatives.restype = ctypes.c_long
        L.smref_ndt_align.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, dp, ctypes.c_float, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ip, ip, dp, dp, ip, dp]
        L.smref_ndt_align.restype = ctypes.c_int
        L._ndt_ready = True
    return L


def _f3(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)[:, :3])
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class NdtGrid:
    """VoxelGridCovariance::applyFilter through the C restatement: key, mean, icov [V,3,3], centroid, valid."""

    def __init__(self, target_f32, resolution=1.0, min_points=6, eig_mult=0.01):
        L = _ndt_lib()
        self._t, pt = _f3(target_f32)
        self._h = L.smref_ndt_grid_build(pt, self._t.shape[0], ctypes.c_float(resolution), min_points, eig_mult)
        n = L.smref_ndt_grid_size(self._h)
        self.key = np.zeros(n, dtype=np.int64); self.mean = np.zeros((n, 3)); self.icov = np.zeros((n, 3, 3))
        self.centroid = np.zeros((n, 3), dtype=np.float32); self.valid = np.zeros(n, dtype=np.int32)
        L.smref_ndt_grid_get(self._h, self.key.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), self.mean.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                             self.icov.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self.centroid.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                             self.valid.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        self.valid = self.valid.astype(bool)

    def compute_derivatives(self, src_f32, trans_f32, pose6, outlier_ratio=0.55, compute_hessian=True, nthreads=1):
        L = _ndt_lib()
        s, ps = _f3(src_f32); t, pt = _f3(trans_f32)
        p, pp = _d(pose6)
        score = ctypes.c_double(); g = np.zeros(6); H = np.zeros((6, 6))
        pairs = L.smref_ndt_compute_derivatives(self._h, ps, pt, s.shape[0], pp, outlier_ratio, int(compute_hessian), nthreads,
                                                ctypes.byref(score), g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return score.value, g, H, int(pairs)

    def __del__(self):
        try:
            if self._h:
                _ndt_lib().smref_ndt_grid_free(self._h)
                self._h = None
        except Exception:
            pass


def ndt_align(source_f32, target_f32, guess=None, resolution=1.0, step_size=0.1, outlier_ratio=0.55, trans_eps=0.1,
              max_iterations=35, nthreads_deriv=6, nthreads_other=1, with_fitness=True):
    """C restatement of Ndt::Align (ndt.cc:38-64 -> pclomp).  nthreads_deriv = 6 is the reference's setting (ndt.cc:32)."""
    L = _ndt_lib()
    s, ps = _f3(source_f32); t, pt = _f3(target_f32)
    g, pg = _d(np.eye(4) if guess is None else guess)
    res = np.zeros((4, 4)); fit = ctypes.c_double(); it = ctypes.c_int(); calls = ctypes.c_int(); tp = ctypes.c_double()
    mn = ctypes.c_double(); nv = ctypes.c_int(); bt = np.zeros(5)
    rc = L.smref_ndt_align(ps, s.shape[0], pt, t.shape[0], pg, ctypes.c_float(resolution), step_size, outlier_ratio, trans_eps,
                           max_iterations, nthreads_deriv, nthreads_other, int(with_fitness),
                           res.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(fit), ctypes.byref(it), ctypes.byref(calls),
                           ctypes.byref(tp), ctypes.byref(mn), ctypes.byref(nv), bt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc != 0:
        raise RuntimeError(f"smref_ndt_align failed: {rc}")
    return dict(result=res, score=fit.value, iterations=it.value, derivative_calls=calls.value, trans_probability=tp.value,
                mean_neighbours=mn.value, voxels=nv.value,
                block_times=dict(applyFilter=bt[0], computeDerivatives=bt[1], transformPointCloud=bt[2], getFitnessScore=bt[3], Align=bt[4]))


# ------------------------------------------------------------------------------------------------------------
# static_map::MultiResolutionVoxelMap (oracle/csrc/smref_mrvm.c)
# ------------------------------------------------------------------------------------------------------------
class Mrvm:
    """The reference's hit / miss voxel map executed in point order (builder/multi_resolution_voxel_map.cc:59-131)."""

    def __init__(self, high_resolution=0.1, hit_prob=0.55, miss_prob=0.48, z_offset=0.0, max_point_num_in_cell=10):
        L = lib()
        L.smref_mrvm_create.restype = ctypes.c_void_p
        L.smref_mrvm_create.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int]
        L.smref_mrvm_free.argtypes = [ctypes.c_void_p]
        L.smref_mrvm_insert.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.smref_mrvm_voxel_count.restype = ctypes.c_long
        L.smref_mrvm_voxel_count.argtypes = [ctypes.c_void_p]
        L.smref_mrvm_dump.restype = ctypes.c_long
        L.smref_mrvm_dump.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.smref_mrvm_output.restype = ctypes.c_long
        L.smref_mrvm_output.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        L.smref_mrvm_output_ex.restype = ctypes.c_long
        L.smref_mrvm_output_ex.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        L.smref_mrvm_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._L = L
        self.max_points = max_point_num_in_cell
        self._h = L.smref_mrvm_create(high_resolution, hit_prob, miss_prob, z_offset, max_point_num_in_cell)
        if not self._h:
            raise ValueError("bad MRVM settings")

    def close(self):
        if self._h:
            self._L.smref_mrvm_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def tables(self):
        hit = np.zeros(256, np.uint8); miss = np.zeros(256, np.uint8)
        self._L.smref_mrvm_tables(self._h, hit.ctypes.data, miss.ctypes.data)
        return hit, miss

    def insert(self, points5, origin):
        p = np.ascontiguousarray(points5, dtype=np.float32)
        assert p.ndim == 2 and p.shape[1] == 5
        o = np.ascontiguousarray(origin, dtype=np.float32)
        return self._L.smref_mrvm_insert(self._h, p.ctypes.data, len(p), o.ctypes.data)

    def dump(self):
        """(keys [V,3], prob [V], max_intensity [V], npoints [V], points [V, max, 5]) sorted by key."""
        n = self._L.smref_mrvm_voxel_count(self._h)
        keys = np.zeros((n, 3), np.int32); prob = np.zeros(n, np.uint8); mi = np.zeros(n, np.int32); npts = np.zeros(n, np.int32)
        pts = np.zeros((n, self.max_points, 5), np.float32)
        self._L.smref_mrvm_dump(self._h, keys.ctypes.data, prob.ctypes.data, mi.ctypes.data, npts.ctypes.data, pts.ctypes.data)
        o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
        return keys[o], prob[o], mi[o], npts[o], pts[o]

    def output(self, threshold=0.6, use_max_intensity=True, average=False, rgb=False):
        flags = (1 if average else 0) | (2 if rgb else 0)
        n = self._L.smref_mrvm_output_ex(self._h, threshold, int(use_max_intensity), flags, None, 0)
        out = np.zeros((n, 4), np.float32)
        self._L.smref_mrvm_output_ex(self._h, threshold, int(use_max_intensity), flags, out.ctypes.data, n)
        return out
