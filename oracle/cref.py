"""ctypes binding of oracle/_build/libsmref.so (C restatement; TEST ORACLE and
timed CPU baseline only -- see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "_build", "libsmref.so")
    if force or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        _LIB.smref_icp_align.argtypes = [dp, ctypes.c_int, dp, dp, ctypes.c_int, dp, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_int, ctypes.c_int, dp, dp, ip, dp, ip, dp]
        _LIB.smref_icp_align.restype = ctypes.c_int
        _LIB.smref_nn.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, ip, dp]
        _LIB.smref_nn.restype = ctypes.c_int
        _LIB.smref_calculate_normals.argtypes = [dp, ctypes.c_int, dp, dp, ip]
        _LIB.smref_calculate_normals.restype = ctypes.c_int
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def icp_fast_align(source, target, target_normals, guess=None, max_iteration=100,
                   dist_outlier_ratio=0.7, early_exit=True, nthreads=1, want_matches=False):
    """C restatement of IcpFast::Align.  Returns dict(result, score, iterations, block_times[, ids, d2])."""
    src, psrc = _d(source)
    tgt, ptgt = _d(target)
    nrm, pnrm = _d(target_normals)
    g, pg = _d(np.eye(4) if guess is None else guess)
    res = np.zeros((4, 4))
    score = ctypes.c_double()
    iters = ctypes.c_int()
    bt = np.zeros(4)
    ids = np.zeros(src.shape[0], dtype=np.int32) if want_matches else None
    d2 = np.zeros(src.shape[0]) if want_matches else None
    rc = lib().smref_icp_align(
        psrc, src.shape[0], ptgt, pnrm, tgt.shape[0], pg, int(max_iteration),
        ctypes.c_float(dist_outlier_ratio), int(early_exit), int(nthreads),
        res.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(score), ctypes.byref(iters),
        bt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if want_matches else None,
        d2.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if want_matches else None)
    if rc != 0:
        raise RuntimeError(f"smref_icp_align failed: {rc}")
    out = dict(result=res, score=score.value, iterations=iters.value,
               block_times=dict(FindClosests=bt[0], ErrorElements=bt[1], ComputePointToPlane=bt[2], BuildKdTree=bt[3]))
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


def calculate_normals(points):
    pts, pp = _d(points)
    n = pts.shape[0]
    op = np.zeros((n, 3)); on = np.zeros((n, 3)); sz = np.zeros(n, dtype=np.int32)
    m = lib().smref_calculate_normals(pp, n, op.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      on.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      sz.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return op[:m].copy(), on[:m].copy(), sz[:m].copy()
