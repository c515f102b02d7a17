"""CPU checks of the NdtWithGicp oracle (oracle/ndt_gicp.py) against itself: no reference golden vectors exist for
this matcher (parity unpinned), so these are the self-consistency known-answer tests SURVEY.md §8(c) asks for."""
import numpy as np

from oracle import ndt_gicp as ong
from staticmapping_amd import synth


def test_voxel_filter_two_formulations_agree():
    rng = np.random.default_rng(0)
    for cloud in (rng.normal(0, 5.0, (20000, 3)), rng.normal(0, 0.3, (3000, 3)), rng.random((50, 3)) * 0.19 - 7.0):
        c = cloud.astype(np.float32)
        a, b = ong.approximate_voxel_grid(c, 0.2), ong.approximate_voxel_grid_runs(c, 0.2)
        assert a.shape == b.shape and np.array_equal(a, b)


def test_voxel_filter_known_answers():
    # two points in one voxel -> their float mean; a third in another voxel with the SAME hash entry evicts it first
    p = np.array([[0.05, 0.05, 0.05], [0.15, 0.05, 0.05]], dtype=np.float32)
    out = ong.approximate_voxel_grid(p, 0.2)
    assert out.shape == (1, 3) and np.array_equal(out[0], (p[0] + p[1]) / np.float32(2))
    # voxel (512, 0, 0) hashes like (0, 0, 0): 512 * 7171 & 511 == 0
    q = np.array([[0.05, 0.05, 0.05], [512 * 0.2 + 0.05, 0.05, 0.05], [0.06, 0.05, 0.05]], dtype=np.float32)
    out = ong.approximate_voxel_grid(q, 0.2)
    assert out.shape == (3, 3)                       # the revisit does not merge with the evicted centroid
    assert np.array_equal(out[0], q[0]) and np.array_equal(out[1], q[1]) and np.array_equal(out[2], q[2])


def test_bfgs_minimises_a_quadratic():
    class Q:
        A = np.diag([1.0, 4.0, 9.0, 2.0, 3.0, 5.0]); b = np.arange(6.0)
        def f(self, x): return float(0.5 * x @ self.A @ x - self.b @ x)
        def df(self, x): return self.A @ x - self.b
        def fdf(self, x): return self.f(x), self.df(x)
    q = Q()
    bf = ong.Bfgs(q)
    x = np.zeros(6)
    bf.init(x)
    for _ in range(60):
        st, x = bf.one_step(x)
        if st or bf.test_gradient(1e-9) == ong.SUCCESS:
            break
    assert np.allclose(x, np.linalg.solve(q.A, q.b), atol=1e-6)


def test_functor_gradient_matches_finite_differences():
    rng = np.random.default_rng(1)
    src = rng.normal(0, 5, (400, 3)).astype(np.float32)
    tgt = (src + rng.normal(0, 0.05, src.shape)).astype(np.float32)
    A = rng.normal(0, 1, (400, 3, 3)); M = A @ np.transpose(A, (0, 2, 1)) + np.eye(3)
    base = synth.make_pose(t=(0.1, -0.2, 0.05), rpy_deg=(1, 2, -3)).astype(np.float32)
    fn = ong.GicpFunctor(base, src, tgt, M)
    x = np.array([0.02, -0.01, 0.03, 0.01, -0.02, 0.015])
    g = fn.df(x)
    # translation part exactly; rotation part uses the reference's own (approximate) R-derivative convention,
    # so only check it against differences of ITS model: f depends on x3..5 through R(x) * base_R
    for k in range(3):
        e = np.zeros(6); e[k] = 1e-3
        fd = (fn.f(x + e) - fn.f(x - e)) / 2e-3
        assert abs(fd - g[k]) < 2e-3 * max(1.0, abs(g[k]))


def test_gicp_recovers_a_known_motion_on_clean_planes():
    rng = np.random.default_rng(2)
    n = 3000
    pl = [np.c_[rng.uniform(-8, 8, n), rng.uniform(-8, 8, n), np.zeros(n)],
          np.c_[rng.uniform(-8, 8, n), np.full(n, 8.0), rng.uniform(0, 6, n)],
          np.c_[np.full(n, -8.0), rng.uniform(-8, 8, n), rng.uniform(0, 6, n)]]
    tgt = np.concatenate(pl).astype(np.float32)
    T = synth.make_pose(t=(0.25, -0.15, 0.1), rpy_deg=(0.5, -0.8, 1.2))
    src = ((tgt.astype(np.float64) - T[:3, 3]) @ T[:3, :3]).astype(np.float32)        # T^-1 applied: T maps src -> tgt
    out = ong.gicp_align(src, tgt, np.eye(4, dtype=np.float32))
    R, t = out["result"][:3, :3].astype(np.float64), out["result"][:3, 3].astype(np.float64)
    ang = np.arccos(np.clip((np.trace(R @ T[:3, :3].T) - 1) / 2, -1, 1))
    assert ang < 2e-3 and np.linalg.norm(t - T[:3, 3]) < 2e-2
    assert out["score"] < 1e-3


def test_gicp_own_repeatability_under_legal_float_roundings(capsys):
    """How far apart do two LEGAL compilations of PCL's GICP land?  The functor transforms every point with a float 4x4
    (gicp_omp_impl.hpp:263-268); whether the compiler contracts the multiply-adds (one rounding) or not (two) moves some
    coordinates by one float ulp.  The Mahalanobis weights reach 1 / gicp_epsilon = 1000 and the line search's Wolfe tests
    (sigma = 0.01) decide on differences of that size, so the two runs part ways after a few BFGS steps.  This is the band a
    GPU implementation (which contracts) can be compared with a CPU build of the reference in -- the tolerance of
    tests/test_ndt_gicp_gpu.py's whole-run check -- measured here on the oracle alone, no GPU involved."""
    scene = synth.make_scene(0)
    tgt_scan = synth.velodyne_scan(scene, synth.make_pose(), seed=71, n_points=30000)
    src_scan = synth.velodyne_scan(scene, synth.make_pose(t=(0.6, 0.05, 0.0), rpy_deg=(0, 0, 1.0)), seed=72, n_points=30000)
    ds = ong.approximate_voxel_grid_runs(src_scan[:, :3], 0.4)
    dt = ong.approximate_voxel_grid_runs(tgt_scan[:, :3], 0.4)
    guess = synth.make_pose(t=(0.45, 0.0, 0.0)).astype(np.float32)
    runs = {mode: ong.gicp_align(ds, dt, guess, transform_mode=mode) for mode in ("nofma", "fma", "blas")}
    truth = synth.make_pose(t=(0.6, 0.05, 0.0), rpy_deg=(0, 0, 1.0))

    def err(A, B):
        R = A[:3, :3].astype(np.float64) @ B[:3, :3].astype(np.float64).T
        return (float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))), float(np.linalg.norm(A[:3, 3].astype(np.float64) - B[:3, 3])))
    lines = []
    worst = (0.0, 0.0)
    for a, b in (("nofma", "fma"), ("nofma", "blas"), ("fma", "blas")):
        da, dt_ = err(runs[a]["result"], runs[b]["result"])
        worst = (max(worst[0], da), max(worst[1], dt_))
        lines.append(f"  {a:5s} vs {b:5s}: {da:.2e} rad {dt_:.2e} m  (iterations {runs[a]['iterations']} / {runs[b]['iterations']})")
    with capsys.disabled():
        print("\n[GICP repeatability under legal float roundings of the functor's transform]")
        print("\n".join(lines))
        for m, r in runs.items():
            print(f"  {m:5s} vs truth: {err(r['result'], truth)[0]:.2e} rad {err(r['result'], truth)[1]:.2e} m  fitness {r['score']:.6f}")
    # every variant is a valid GICP result (close to the truth, similar fitness) ...
    for r in runs.values():
        da, dt_ = err(r["result"], truth)
        assert da < 1e-2 and dt_ < 0.25 and r["score"] < 0.1
    # ... and two of them do NOT agree with each other to the north star's 1e-3 m (measured: 3.7e-4 rad / 0.14 m between
    # the contracted and the uncontracted build on this pair): whole-run GICP parity beyond its own repeatability is not a
    # property the reference has, which is why tests/test_ndt_gicp_gpu.py pins the stages and bounds the whole run loosely
    assert 1e-3 < worst[1] < 0.3 and worst[0] < 5e-3
