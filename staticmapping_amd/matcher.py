"""Python view of the IcpFast-equivalent matcher; mirrors registrator::Interface
(/root/reference/registrators/interface.h:67-116): SetInputSource / SetInputTarget /
Align / GetFitnessScore, with 4x4 numpy matrices in the usual (row, col) indexing.
All compute happens inside libsmhip.so on the GPU."""
from __future__ import annotations

import ctypes
import numpy as np

from . import _capi

NN_BRUTE = 0
NN_GRID = 1
NN_NABO = 2          # libnabo's tree and epsilon-approximate search (the reference's own semantics)


class SmhipError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"smhip status {status}: {msg}")
        self.status = status


def se3_error(Ta: np.ndarray, Tb: np.ndarray):
    """(rotation angle of Ra Rb^T [rad], |ta - tb| [m]) -- the metric of BASELINE.json."""
    R = Ta[:3, :3] @ Tb[:3, :3].T
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    s = np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2.0
    return float(np.arctan2(s, c)), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def calculate_normals(points):
    """Host-side EigenPointCloud::CalculateNormals (cloud_types.cc:347-368) from libsmhip.so:
    returns (kept_points [M,3], normals [M,3]).  Leaves whose normal is not finite (singular
    M, which the reference does not check) are dropped here so IcpFast never sees NaN."""
    lib = _capi.load_library()
    p = _f64(np.asarray(points)[:, :3])
    n = p.shape[0]
    op = np.zeros((n, 3)); on = np.zeros((n, 3)); m = ctypes.c_int32()
    st = lib.smhip_calculate_normals_f64(p.ctypes.data_as(_capi.c_double_p), n, op.ctypes.data_as(_capi.c_double_p),
                                         on.ctypes.data_as(_capi.c_double_p), ctypes.byref(m))
    if st != 0:
        raise SmhipError(st, lib.smhip_status_string(st).decode())
    op, on = op[:m.value], on[:m.value]
    ok = np.isfinite(on).all(axis=1)
    return op[ok].copy(), on[ok].copy()


class IcpFastHip:
    """One matcher instance = one smhip handle with `pair_slots` independent scan pairs.

    Slot 0 plays the role of the reference's single source/target pair; the other
    slots exist for the batched / sharded throughput path.
    """

    def __init__(self, device: int = 0, pair_slots: int = 1, max_source_points: int = 131072,
                 max_target_points: int = 131072, stream: int | None = None, **options):
        self._lib = _capi.load_library()
        self._h = ctypes.c_void_p()
        st = self._lib.smhip_create(device, ctypes.c_void_p(stream) if stream else None, pair_slots,
                                    max_source_points, max_target_points, ctypes.byref(self._h))
        if st != 0:
            self._h = ctypes.c_void_p()
            raise SmhipError(st, self._lib.smhip_status_string(st).decode())
        self.pair_slots = pair_slots
        self.final_score_ = float("nan")
        self.last_stats = None
        self._opts = _capi.IcpOptions()
        self._lib.smhip_icp_default_options(ctypes.byref(self._opts))
        if options:
            self.set_options(**options)

    # -- lifetime --------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.smhip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != 0:
            raise SmhipError(st, self._lib.smhip_last_error(self._h).decode())

    # -- options (names of icp_fast.cc:407-419 + backend knobs) ----------
    def set_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self._opts, k):
                raise KeyError(f"unknown option {k}")     # interface.cc:66-67 CHECK on unknown names
            setattr(self._opts, k, v)
        self._check(self._lib.smhip_icp_set_options(self._h, ctypes.byref(self._opts)))

    # -- inputs ----------------------------------------------------------
    def set_input_source(self, points, slot: int = 0):
        """points: [N,3+] array; float32 arrays are uploaded as-is (InnerPointType / KITTI rows),
        anything else goes through the float64 EigenPointCloud entry point."""
        a = np.asarray(points)
        if a.dtype == np.float32 and a.ndim == 2 and a.flags.c_contiguous:
            self._check(self._lib.smhip_set_source_f32(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0]))
        else:
            a = _f64(a[:, :3])
            self._check(self._lib.smhip_set_source_f64(self._h, slot, a.ctypes.data_as(_capi.c_double_p), a.shape[0]))

    def set_input_sources_batch(self, clouds, slots):
        """SetInputSource of many pair slots in one call (smhip_set_sources_f32_batch): clouds = float32 [N, 4] arrays (KITTI
        rows); one Morton ordering for the whole batch.  Returns once the copies are enqueued; numpy arrays are pageable, so
        they are staged one at a time (page-locked buffers are copied from where they lie)."""
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        for a in arrs:
            if a.ndim != 2 or a.shape[1] != 4:
                raise ValueError("the batched upload takes rows of 4 floats (x y z reflectance)")
        k = len(arrs)
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        n = np.ascontiguousarray([len(a) for a in arrs], dtype=np.int32)
        rows = (_capi.c_float_p * k)(*[a.ctypes.data_as(_capi.c_float_p) for a in arrs])
        self._check(self._lib.smhip_set_sources_f32_batch(self._h, k, sl.ctypes.data_as(_capi.c_int32_p), rows, n.ctypes.data_as(_capi.c_int32_p)))
        self.synchronize()

    def set_input_target(self, points, normals=None, slot: int = 0):
        p = _f64(np.asarray(points)[:, :3])
        n = None if normals is None else _f64(np.asarray(normals)[:, :3])
        self._check(self._lib.smhip_set_target_f64(
            self._h, slot, p.ctypes.data_as(_capi.c_double_p),
            n.ctypes.data_as(_capi.c_double_p) if n is not None else None, p.shape[0]))

    def prepare_target(self, scan, slot: int = 0) -> int:
        """Upload a raw float32 scan [N,3+] and run CalculateNormals on the GPU; returns the target size."""
        a = np.ascontiguousarray(np.asarray(scan, dtype=np.float32))
        m = ctypes.c_int32()
        self._check(self._lib.smhip_prepare_target_f32(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0], ctypes.byref(m)))
        return m.value

    def prepare_target_from_source(self, from_slot: int, to_slot: int) -> int:
        m = ctypes.c_int32()
        self._check(self._lib.smhip_prepare_target_from_source(self._h, from_slot, to_slot, ctypes.byref(m)))
        return m.value

    def prepare_targets_from_sources(self, from_slots, to_slots):
        """Batched device CalculateNormals: target of to_slots[k] from the source cloud resident in from_slots[k]."""
        f = np.ascontiguousarray(from_slots, dtype=np.int32); t = np.ascontiguousarray(to_slots, dtype=np.int32)
        m = np.zeros(len(f), dtype=np.int32)
        self._check(self._lib.smhip_prepare_targets_from_sources(self._h, len(f), f.ctypes.data_as(_capi.c_int32_p),
                                                                 t.ctypes.data_as(_capi.c_int32_p), m.ctypes.data_as(_capi.c_int32_p)))
        return m

    def get_target(self, n: int, slot: int = 0):
        p = np.zeros((n, 3), np.float32); nr = np.zeros((n, 3), np.float32)
        self._check(self._lib.smhip_get_target_f32(self._h, slot, p.ctypes.data_as(_capi.c_float_p), nr.ctypes.data_as(_capi.c_float_p), n))
        return p, nr

    def set_target_cache(self, enable: bool = True):
        """Keep the target-side structures (ICP search grid, NDT voxel table) across single-pair calls while the target is
        unchanged (default on); off = rebuild on every Align like the reference.  Results are identical either way."""
        self._check(self._lib.smhip_set_target_cache(self._h, 1 if enable else 0))

    def forget_search_history(self):
        """The next batch chooses where to switch search forms as a new handle's first batch does (smhip_icp_forget_search_history)."""
        self._check(self._lib.smhip_icp_forget_search_history(self._h))

    def single_launch_counts(self):
        """(Aligns that ran as one cooperative launch, times such a launch stopped itself and the Align was redone as separate
        launches) -- smhip_icp_single_launch_counts."""
        import ctypes
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.smhip_icp_single_launch_counts(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def copy_slot(self, src_slot: int, dst_slot: int):
        self._check(self._lib.smhip_copy_slot(self._h, src_slot, dst_slot))

    # -- Align -----------------------------------------------------------
    def align(self, guess=None):
        """IcpFast::Align for slot 0.  Returns (ok, result 4x4)."""
        res, scores, stats = self.align_batch(1, None if guess is None else [guess])
        self.final_score_ = float(scores[0])
        return True, res[0]

    def get_fitness_score(self) -> float:
        return self.final_score_

    @staticmethod
    def _pack_guesses(npairs, guesses):
        g = np.empty((npairs, 16), dtype=np.float64)
        for p in range(npairs):
            G = np.eye(4) if guesses is None else np.asarray(guesses[p], dtype=np.float64)
            g[p] = G.T.reshape(-1)          # column-major
        return g

    def align_batch(self, npairs: int, guesses=None):
        g = self._pack_guesses(npairs, guesses)
        res = np.zeros((npairs, 16))
        scores = np.zeros(npairs)
        stats = (_capi.IcpStats * npairs)()
        st = self._lib.smhip_icp_align_batch(self._h, npairs, g.ctypes.data_as(_capi.c_double_p),
                                             res.ctypes.data_as(_capi.c_double_p),
                                             scores.ctypes.data_as(_capi.c_double_p), stats)
        self.last_stats = [dict(iterations=s.iterations, kept=s.kept, limit_d2=s.limit_d2,
                                fallback_queries=s.fallback_queries, status=s.status,
                                hard_queries=s.hard_queries, refined_iterations=s.refined_iterations,
                                searched_queries=s.searched_queries, fused_iterations=s.fused_iterations) for s in stats]
        self._check(st)
        return res.reshape(npairs, 4, 4).transpose(0, 2, 1).copy(), scores, self.last_stats

    def enqueue_batch(self, npairs: int, guesses=None):
        g = self._pack_guesses(npairs, guesses)
        self._check(self._lib.smhip_icp_enqueue_batch(self._h, npairs, g.ctypes.data_as(_capi.c_double_p)))

    def fetch_batch(self, npairs: int):
        res = np.zeros((npairs, 16))
        scores = np.zeros(npairs)
        stats = (_capi.IcpStats * npairs)()
        st = self._lib.smhip_icp_fetch_batch(self._h, npairs, res.ctypes.data_as(_capi.c_double_p),
                                             scores.ctypes.data_as(_capi.c_double_p), stats)
        self.last_stats = [dict(iterations=s.iterations, kept=s.kept, limit_d2=s.limit_d2,
                                fallback_queries=s.fallback_queries, status=s.status,
                                hard_queries=s.hard_queries, refined_iterations=s.refined_iterations,
                                searched_queries=s.searched_queries, fused_iterations=s.fused_iterations) for s in stats]
        self._check(st)
        return res.reshape(npairs, 4, 4).transpose(0, 2, 1).copy(), scores, self.last_stats

    def synchronize(self):
        self._check(self._lib.smhip_synchronize(self._h))

    # -- introspection ---------------------------------------------------
    def get_matches(self, n: int, slot: int = 0):
        ids = np.zeros(n, dtype=np.int32)
        d2 = np.zeros(n, dtype=np.float32)
        self._check(self._lib.smhip_icp_get_matches(self._h, slot, ids.ctypes.data_as(_capi.c_int32_p),
                                                    d2.ctypes.data_as(_capi.c_float_p), n))
        return ids, d2

    def search_counts(self, slot: int = 0):
        """Queries that went through a search in iterations 0..11 of the slot's last Align."""
        c = (ctypes.c_uint32 * 12)()
        self._check(self._lib.smhip_icp_get_search_counts(self._h, slot, c))
        return list(c)

    def find_closests(self, T, n: int):
        Tc = _f64(np.asarray(T).T).reshape(-1)
        ids = np.zeros(n, dtype=np.int32)
        d2 = np.zeros(n, dtype=np.float32)
        self._check(self._lib.smhip_icp_find_closests(self._h, 0, Tc.ctypes.data_as(_capi.c_double_p),
                                                      ids.ctypes.data_as(_capi.c_int32_p),
                                                      d2.ctypes.data_as(_capi.c_float_p), n))
        return ids, d2

    def export_results_device(self, npairs: int, dev_ptr: int):
        """Write npairs x 18 doubles (16 col-major transform, score, iterations) to device memory."""
        self._check(self._lib.smhip_icp_export_results_device(self._h, npairs, ctypes.c_void_p(dev_ptr)))

    def enable_profile(self, on=True):
        """False/0 off, True/1 HIP events around every launch; one kernel class only (cheap enough for a timed region):
        2 the NN kernels proper, 3 accumulate, 4 the listed search."""
        self._check(self._lib.smhip_icp_enable_profile(self._h, int(on)))

    def get_profile(self) -> dict:
        p = _capi.IcpProfile()
        self._check(self._lib.smhip_icp_get_profile(self._h, ctypes.byref(p)))
        return {k: getattr(p, k) for k, _ in p._fields_ if k != "reserved"}


class NdtHip(IcpFastHip):
    """registrators::Ndt (pclomp NDT) on the GPU: mirrors Ndt::Align (/root/reference/registrators/ndt.cc:38-64).
    Clouds are float32 [N,3+] arrays (the reference converts InnerPointType -> pcl::PointXYZ); the
    target needs no normals.  get_fitness_score() is PCL's getFitnessScore(): mean squared 1-NN
    distance, LOWER is better."""

    def __init__(self, device: int = 0, max_source_points: int = 131072, max_target_points: int = 524288,
                 stream: int | None = None, pair_slots: int = 1, **ndt_options):
        super().__init__(device=device, pair_slots=pair_slots, max_source_points=max_source_points,
                         max_target_points=max_target_points, stream=stream)
        self._nopts = _capi.NdtOptions()
        self._lib.smhip_ndt_default_options(ctypes.byref(self._nopts))
        if ndt_options:
            self.set_ndt_options(**ndt_options)
        self.last_ndt_stats = None

    def set_ndt_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self._nopts, k):
                raise KeyError(f"unknown option {k}")
            setattr(self._nopts, k, v)
        self._check(self._lib.smhip_ndt_set_options(self._h, ctypes.byref(self._nopts)))

    def set_input_source(self, points, slot: int = 0):
        a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
        self._check(self._lib.smhip_set_source_f32(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0]))

    def set_input_target(self, points, normals=None, slot: int = 0):
        a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
        self._check(self._lib.smhip_set_target_f32(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], None, 0, a.shape[0]))

    def align(self, guess=None):
        G = np.eye(4) if guess is None else np.asarray(guess, dtype=np.float64)
        g = np.ascontiguousarray(G.T).reshape(-1)
        res = np.zeros(16)
        score = ctypes.c_double()
        st = _capi.NdtStats()
        self._check(self._lib.smhip_ndt_align(self._h, g.ctypes.data_as(_capi.c_double_p), res.ctypes.data_as(_capi.c_double_p),
                                              ctypes.byref(score), ctypes.byref(st)))
        self.final_score_ = score.value
        self.last_ndt_stats = {k: getattr(st, k) for k, _ in st._fields_}
        return True, res.reshape(4, 4).T.copy()

    def align_batch(self, npairs: int, guesses=None, first_slot: int = 0):
        """npairs independent Ndt::Align calls on slots first_slot .. first_slot + npairs - 1, advanced in lock-step
        (smhip_ndt_align_batch): poses [npairs, 4, 4], fitness scores [npairs], stats (list of dicts) -- the single calls' bits."""
        g = self._pack_guesses(npairs, guesses)
        res = np.zeros((npairs, 16))
        scores = np.zeros(npairs)
        stats = (_capi.NdtStats * npairs)()
        self._check(self._lib.smhip_ndt_align_batch(self._h, first_slot, npairs, g.ctypes.data_as(_capi.c_double_p), res.ctypes.data_as(_capi.c_double_p),
                                                    scores.ctypes.data_as(_capi.c_double_p), stats))
        self.last_ndt_stats = [{k: getattr(s, k) for k, _ in s._fields_} for s in stats]
        return res.reshape(npairs, 4, 4).transpose(0, 2, 1).copy(), scores, self.last_ndt_stats

    # -- parity-test hooks ---------------------------------------------------------------------
    def build_voxels(self) -> int:
        n = ctypes.c_int()
        self._check(self._lib.smhip_ndt_build_voxels(self._h, ctypes.byref(n)))
        return n.value

    def get_voxels(self, n: int):
        keys = np.zeros(n, np.int32); counts = np.zeros(n, np.int32)
        means = np.zeros((n, 3)); icov = np.zeros((n, 6), np.float32); cent = np.zeros((n, 3), np.float32)
        self._check(self._lib.smhip_ndt_get_voxels(self._h, n, keys.ctypes.data_as(_capi.c_int32_p), counts.ctypes.data_as(_capi.c_int32_p),
                                                   means.ctypes.data_as(_capi.c_double_p), icov.ctypes.data_as(_capi.c_float_p),
                                                   cent.ctypes.data_as(_capi.c_float_p)))
        return keys, counts, means, icov, cent

    def compute_derivatives(self, pose6, compute_hessian: bool = True):
        p = _f64(pose6)
        score = ctypes.c_double(); g = np.zeros(6); H = np.zeros(36)
        self._check(self._lib.smhip_ndt_compute_derivatives(self._h, p.ctypes.data_as(_capi.c_double_p), int(compute_hessian),
                                                            ctypes.byref(score), g.ctypes.data_as(_capi.c_double_p),
                                                            H.ctypes.data_as(_capi.c_double_p)))
        return score.value, g, H.reshape(6, 6)

    def time_derivatives(self, npairs: int = 1, launches: int = 20, first_slot: int = 0):
        """HIP-event time of back-to-back computeDerivatives launches over the slots' last poses (smhip_ndt_time_derivatives):
        (ms per launch, (point, voxel) pairs per launch)."""
        ms = np.zeros(1); pairs = np.zeros(1)
        self._check(self._lib.smhip_ndt_time_derivatives(self._h, first_slot, npairs, launches, ms.ctypes.data_as(_capi.c_double_p),
                                                         pairs.ctypes.data_as(_capi.c_double_p)))
        return float(ms[0]), float(pairs[0])


class NdtGicpHip(IcpFastHip):
    """registrators::NdtWithGicp on the GPU: mirrors NdtWithGicp::Align
    (/root/reference/registrators/ndt_gicp.cc:55-112): ApproximateVoxelGrid(0.2 m) on both clouds -> pcl NDT ->
    pcl GICP; get_fitness_score() = exp(-GICP fitness), HIGHER is better.  Clouds are float32 [N,3+] arrays.
    Options carry the reference's names (`use_ndt`, `using_voxel_filter`, `voxel_resolution`, ndt_gicp.cc:31-36)
    plus the PCL parameters the reference fixes in InitWithOptions (:45-53)."""

    def __init__(self, device: int = 0, max_source_points: int = 131072, max_target_points: int = 524288,
                 stream: int | None = None, jobs: int = 1, **options):
        # a job = one (source, target) pair with its own kept structures; job j works in pair slot j and slot jobs + j
        super().__init__(device=device, pair_slots=2 * max(1, jobs), max_source_points=max_source_points,
                         max_target_points=max(max_target_points, max_source_points), stream=stream)
        self.jobs = max(1, jobs)
        self._gopts = _capi.NdtGicpOptions()
        self._lib.smhip_ndt_gicp_default_options(ctypes.byref(self._gopts))
        if options:
            self.set_gicp_options(**options)
        self.last_gicp_stats = None

    def set_gicp_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self._gopts, k):
                raise KeyError(f"unknown option {k}")
            setattr(self._gopts, k, v)
        self._check(self._lib.smhip_ndt_gicp_set_options(self._h, ctypes.byref(self._gopts)))

    def set_input_source(self, points, slot: int = 0):
        """`slot` = the job (0 for the single Align)."""
        a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
        self._check(self._lib.smhip_ndt_gicp_set_source_f32_job(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0]))

    def set_input_target(self, points, normals=None, slot: int = 0):
        a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
        self._check(self._lib.smhip_ndt_gicp_set_target_f32_job(self._h, slot, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0]))

    def _run(self, fn, guess):
        G = np.eye(4) if guess is None else np.asarray(guess, dtype=np.float64)
        g = np.ascontiguousarray(G.T).reshape(-1)
        res = np.zeros(16)
        score = ctypes.c_double()
        st = _capi.NdtGicpStats()
        self._check(fn(self._h, g.ctypes.data_as(_capi.c_double_p), res.ctypes.data_as(_capi.c_double_p),
                       ctypes.byref(score), ctypes.byref(st)))
        self.last_gicp_stats = {k: getattr(st, k) for k, _ in st._fields_ if k != "reserved"}
        return score.value, res.reshape(4, 4).T.copy()

    def align(self, guess=None):
        self.final_score_, result = self._run(self._lib.smhip_ndt_gicp_align, guess)
        return bool(self.last_gicp_stats["ok"]), result

    def align_batch(self, njobs: int, guesses=None, first_job: int = 0):
        """njobs NdtWithGicp::Align calls (jobs first_job .. first_job + njobs - 1) advanced in lock-step
        (smhip_ndt_gicp_align_batch): poses [njobs, 4, 4], scores [njobs], stats (list of dicts) -- the single calls' bits."""
        g = self._pack_guesses(njobs, guesses)
        res = np.zeros((njobs, 16))
        scores = np.zeros(njobs)
        stats = (_capi.NdtGicpStats * njobs)()
        self._check(self._lib.smhip_ndt_gicp_align_batch(self._h, first_job, njobs, g.ctypes.data_as(_capi.c_double_p), res.ctypes.data_as(_capi.c_double_p),
                                                         scores.ctypes.data_as(_capi.c_double_p), stats))
        self.last_gicp_stats = [{k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"} for s in stats]
        return res.reshape(njobs, 4, 4).transpose(0, 2, 1).copy(), scores, self.last_gicp_stats

    # -- parity-test hooks ---------------------------------------------------------------------
    def gicp_only(self, source, target, guess=None):
        """pcl GICP alone (no filter, no NDT) on the given clouds; returns (fitness, result)."""
        a = np.ascontiguousarray(np.asarray(source, dtype=np.float32))
        b = np.ascontiguousarray(np.asarray(target, dtype=np.float32))
        self._check(self._lib.smhip_set_source_f32(self._h, 0, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0]))
        self._check(self._lib.smhip_set_target_f32(self._h, 0, b.ctypes.data_as(_capi.c_float_p), b.shape[1], None, 0, b.shape[0]))
        return self._run(self._lib.smhip_gicp_align, guess)

    def gicp_evaluate(self, guess, x):
        """The GICP functor (f, gradient[6]) at state x over the last run's final correspondences."""
        g = np.ascontiguousarray(np.asarray(guess, dtype=np.float64).T).reshape(-1)
        xx = _f64(x)
        f = ctypes.c_double(); grad = np.zeros(6)
        self._check(self._lib.smhip_gicp_evaluate(self._h, g.ctypes.data_as(_capi.c_double_p), xx.ctypes.data_as(_capi.c_double_p),
                                                  ctypes.byref(f), grad.ctypes.data_as(_capi.c_double_p)))
        return f.value, grad

    def get_downsampled(self, which: int) -> np.ndarray:
        n = ctypes.c_int()
        self._check(self._lib.smhip_ndt_gicp_get_downsampled(self._h, which, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 3), np.float32)
        self._check(self._lib.smhip_ndt_gicp_get_downsampled(self._h, which, out.ctypes.data_as(_capi.c_float_p), n.value, ctypes.byref(n)))
        return out

    def get_covariances(self, which: int, n: int) -> np.ndarray:
        c = np.zeros((n, 6))
        self._check(self._lib.smhip_gicp_get_covariances(self._h, which, c.ctypes.data_as(_capi.c_double_p), n))
        C = np.zeros((n, 3, 3))
        C[:, 0, 0] = c[:, 0]; C[:, 0, 1] = C[:, 1, 0] = c[:, 1]; C[:, 0, 2] = C[:, 2, 0] = c[:, 2]
        C[:, 1, 1] = c[:, 3]; C[:, 1, 2] = C[:, 2, 1] = c[:, 4]; C[:, 2, 2] = c[:, 5]
        return C


class IcpPointMatcherHip:
    """registrators::IcpUsingPointMatcher (/root/reference/registrators/icp_pointmatcher.cc:104-247) on the
    same GPU engine: IcpFast is the author's restatement of exactly this libpointmatcher chain.

      reading filter   RandomSampling(prob 0.9)                       :171-176   (seeded here; std::rand() there)
      reference filter SamplingSurfaceNormal(knn 7, method 1)         :178-184   = CalculateNormals
      matcher          KDTree knn 1 (exact here, eps 3.16 there)      :187-193
      outlier filter   TrimmedDist(0.7)                               :196-200
      minimiser        PointToPlane                                   :207-209
      checkers         Counter(150) + Differential(1e-3, 1e-2, 4)     :212-224
    After compute() the score is recomputed by matching the FULL transformed reading against the RAW
    reference with the same trimmed filter (:112-143) and Align returns score >= 0.6 (:145-148).
    """

    def __init__(self, device: int = 0, max_points: int = 262144, prob: float = 0.9, seed: int | None = 0,
                 device_chain: bool = True, **options):
        # slot 0: the ICP pair (sampled reading vs CalculateNormals(reference)); slot 1: the score pair (full reading vs
        # raw reference).  Each cloud goes up once; sampling, normals and the score pass stay on the device.
        self._m = IcpFastHip(device=device, pair_slots=2, max_source_points=max_points, max_target_points=max_points, **options)
        self.prob = prob
        self.seed = seed
        self.device_chain = device_chain
        self.final_score_ = float("nan")
        self._reading = None
        self._reference = None
        self._reading_on_device = False      # slot 1 holds the current reading / raw reference (uploaded once per cloud,
        self._reference_on_device = False    # like IcpPointMatcherHip in include/smhip/registrator.h)
        self._prepared = None                # (seed, prob) the sampled reading in slot 0 was drawn with
        self._target_prepared = False        # slot 0's target is CalculateNormals of the current reference
        self.last_mask = None

    def close(self):
        self._m.close()

    def set_input_source(self, points):
        a = np.asarray(points, dtype=np.float32)
        self._reading = a[~np.isnan(a[:, :3]).any(axis=1)]          # InnerCloudToPmPoints drops NaN points, :57-66
        self._reading_on_device = False
        self._prepared = None

    def set_input_target(self, points):
        a = np.asarray(points, dtype=np.float32)
        self._reference = a[~np.isnan(a[:, :3]).any(axis=1)]
        self._reference_on_device = False
        self._target_prepared = False

    def sampling_mask(self, n: int) -> np.ndarray:
        """The kept set of the device sampler (smhip_sample_source): uniform = splitmix64(seed << 32 | i) >> 11 * 2^-53,
        kept when < prob.  A pure function of (seed, row), so the oracle can be handed the same mask."""
        if self.prob >= 1.0:
            return np.ones(n, dtype=bool)
        seed = np.uint64((0 if self.seed is None else int(self.seed)) & 0xffffffff)
        with np.errstate(over="ignore"):
            z = (seed << np.uint64(32) | np.arange(n, dtype=np.uint64)) + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return u < np.float64(np.float32(self.prob))

    def align(self, guess=None):
        if self._reading is None or self._reference is None:
            raise SmhipError(4, "Align before SetInputSource/SetInputTarget")
        m = self._m
        lib, h = m._lib, m._h
        G = np.eye(4) if guess is None else np.asarray(guess, dtype=np.float64)
        self.last_mask = self.sampling_mask(len(self._reading))
        m.set_options(max_iteration=150, dist_outlier_ratio=0.7, early_exit=1)
        if self.device_chain:
            # each cloud goes up once: repeated align() calls on the same clouds skip the uploads, the sampling and the
            # target preparation, so the slots' kept search structures (smhip_set_target_cache) are actually reused
            if not self._reading_on_device:
                m.set_input_source(np.ascontiguousarray(self._reading), slot=1)
                self._reading_on_device = True
            if not self._reference_on_device:
                m.set_input_target(np.ascontiguousarray(self._reference[:, :3]), None, slot=1)
                self._reference_on_device = True
            n_out = ctypes.c_int32()
            seed = (0 if self.seed is None else int(self.seed)) & 0xffffffff
            if self._prepared != (seed, float(self.prob)):
                m._check(lib.smhip_sample_source(h, 1, 0, float(self.prob), seed, ctypes.byref(n_out)))  # reading filter :170-174
                if n_out.value != int(self.last_mask.sum()):
                    raise SmhipError(5, f"device sampler kept {n_out.value} points, the host mask {int(self.last_mask.sum())}")
                self._prepared = (seed, float(self.prob))
            if not self._target_prepared:
                m._check(lib.smhip_prepare_target_from_target(h, 1, 0, ctypes.byref(n_out)))              # reference filter :176-184
                self.target_points = n_out.value
                self._target_prepared = True
            # ---- pm_icp_.compute(reading, reference, guess)                            :107-110
            _, result = m.align(G)
            self.iterations = m.last_stats[0]["iterations"]
            # ---- final score: full reading, raw reference, one trimmed matching pass    :112-143
            Tc = np.asfortranarray(result, dtype=np.float64)
            score = ctypes.c_double()
            kept = ctypes.c_int32()
            m._check(lib.smhip_icp_trimmed_score(h, 1, Tc.ctypes.data_as(_capi.c_double_p), 0.7, ctypes.byref(score), ctypes.byref(kept)))
            self.final_score_ = score.value
            self.score_kept = kept.value
            return self.final_score_ >= 0.6, result                                    # :145-148
        # host-side chain (kept for A/B tests of the device chain): host CalculateNormals, four uploads, a second Align
        self._reading_on_device = self._reference_on_device = self._target_prepared = False    # this path overwrites slot 0 / 1
        self._prepared = None
        q, n = calculate_normals(self._reference[:, :3].astype(np.float64))
        m.set_input_source(np.ascontiguousarray(self._reading[self.last_mask]))
        m.set_input_target(q, n)
        _, result = m.align(G)
        self.iterations = m.last_stats[0]["iterations"]
        m.set_options(max_iteration=1, early_exit=0)
        m.set_input_source(np.ascontiguousarray(self._reading))
        ref = self._reference[:, :3].astype(np.float64)
        m.set_input_target(ref, np.zeros_like(ref) + [0.0, 0.0, 1.0])     # normals unused by the score
        m.align(result)
        self.final_score_ = m.get_fitness_score()
        return self.final_score_ >= 0.6, result                                        # :145-148

    def get_fitness_score(self) -> float:
        return self.final_score_
