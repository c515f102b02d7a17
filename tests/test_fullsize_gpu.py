"""Parity at the sizes BASELINE.json quotes, through the C ABI, against the oracle:
  config #3  registrators::Ndt          120k-pt scan vs 500k-pt submap, 1.0 m voxels
  config #5  registrators::NdtWithGicp  120k-pt scan vs 2M-pt submap: the deterministic stages (voxel filter bit-exact,
             NDT stage 1e-4 rad / 1e-3 m, GICP covariances); the BFGS end result is covered at 30k points in
             tests/test_ndt_gicp_gpu.py to GICP's own repeatability
  config #2  IcpFast with the two target variants of SURVEY §8(d): CalculateNormals target [faithful] and per-point
             normals on the full 120k-pt target [stress]
The oracle legs are the C restatements (oracle/csrc), which finish in seconds at these sizes."""
import numpy as np
import pytest

import staticmapping_amd as sm
from staticmapping_amd import synth

pytestmark = pytest.mark.gpu


def _submap(n_scans, n_target, seed):
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.4 * k)) for k in range(n_scans + 1)]
    import torch
    dev = torch.device("cuda", 0)
    scans = [synth.velodyne_scan(scene, P, seed=seed * 50 + k, n_points=120_000, device=dev) for k, P in enumerate(poses)]
    tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:n_scans], poses[:n_scans])])
    rng = np.random.default_rng(seed)
    tgt = np.ascontiguousarray(tgt[np.sort(rng.choice(len(tgt), size=n_target, replace=False))].astype(np.float32))
    T = poses[n_scans]
    G = T.copy(); G[0, 3] -= 0.3
    c, s = np.cos(np.deg2rad(1.0)), np.sin(np.deg2rad(1.0))
    G[:3, :3] = T[:3, :3] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    return np.ascontiguousarray(scans[n_scans][:, :3]), tgt, T, G


@pytest.fixture(scope="module")
def cfg3():
    return _submap(5, 500_000, 4)


def test_config3_ndt_voxels_derivatives_and_align(cfg3):
    from oracle import cref
    from oracle import ndt as ondt
    src, tgt, T, G = cfg3
    m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
    m.set_input_source(src); m.set_input_target(tgt)
    # ---- a13: the voxel table
    grid = cref.NdtGrid(tgt)
    n = m.build_voxels()
    keys, counts, means, icov, cent = m.get_voxels(n)
    searchable = (counts >= 6) | (counts == -1)
    got = dict(zip(keys[searchable].tolist(), np.nonzero(searchable)[0].tolist()))
    assert sorted(got) == sorted(grid.key.tolist())
    idx = np.array([got[k] for k in grid.key.tolist()])
    assert np.allclose(means[idx], grid.mean, rtol=0, atol=1e-9)
    ic = grid.icov
    ref6 = np.stack([ic[:, 0, 0], ic[:, 0, 1], ic[:, 0, 2], ic[:, 1, 1], ic[:, 1, 2], ic[:, 2, 2]], axis=1)
    ok = grid.valid
    assert ((counts[idx] == -1) == ~ok).all()
    scale = np.abs(ref6[ok]).max(axis=1, keepdims=True)
    assert (np.abs(icov[idx][ok] - ref6[ok]) <= 5e-5 * scale + 1e-6).all()
    # ---- a14 / a15: one evaluation
    p = np.zeros(6); p[:3] = G[:3, 3]; p[3:] = ondt.euler_xyz_from_matrix(G[:3, :3])
    tr = ondt.transform_cloud_f32(src, ondt.pose_to_matrix_f32(p))
    s_o, g_o, H_o, n_o = grid.compute_derivatives(src, tr, p, nthreads=cref.usable_cores())
    s_g, g_g, H_g = m.compute_derivatives(p, True)
    assert abs(s_g - s_o) <= 1e-4 * abs(s_o)
    assert np.allclose(g_g, g_o, rtol=2e-3, atol=1e-3 * np.abs(g_o).max())
    assert np.allclose(H_g, H_o, rtol=2e-3, atol=1e-3 * np.abs(H_o).max())
    # ---- a16 / a17: the whole Align + getFitnessScore
    ok_, R = m.align(G)
    ref = cref.ndt_align(src, tgt, guess=G, nthreads_deriv=cref.usable_cores(), nthreads_other=cref.usable_cores())
    st = m.last_ndt_stats
    assert st["iterations"] == ref["iterations"] and st["derivative_calls"] == ref["derivative_calls"]
    assert st["voxels"] >= ref["voxels"]                      # the device table also lists voxels with < 6 points
    da, dt = sm.se3_error(R, ref["result"])
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(m.get_fitness_score() - ref["score"]) <= 1e-3 * ref["score"]
    assert abs(st["trans_probability"] - ref["trans_probability"]) <= 1e-4 * abs(ref["trans_probability"])
    print(f"config #3: {st}, vs oracle {da:.2e} rad {dt:.2e} m, vs truth {sm.se3_error(R, T)}")
    m.close()


def test_config5_ndt_gicp_stages():
    from oracle import ndt_gicp as ong
    from oracle import ndt as ondt
    from scipy.spatial import cKDTree
    src, tgt, T, G = _submap(20, 2_000_000, 6)
    m = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
    m.set_input_source(src); m.set_input_target(tgt)
    # NDT stage alone: no correspondence inside the gate -> GICP stops at once and hands NDT's pose back
    m.set_gicp_options(gicp_corr_dist_threshold=1e-9)
    ok, R = m.align(G)
    st = dict(m.last_gicp_stats)
    # ---- ApproximateVoxelGrid: bit-exact, order included, on both clouds (2M points through the literal serial loop)
    ds, dt_ = m.get_downsampled(0), m.get_downsampled(1)
    os_ = ong.approximate_voxel_grid(src, 0.2)
    ot = ong.approximate_voxel_grid(tgt, 0.2)
    assert np.array_equal(ds, os_) and np.array_equal(dt_, ot)
    # ---- stock-PCL NDT stage (double inner math) on the filtered clouds
    ref = ondt.ndt_align(os_, ot, guess=G, trans_eps=0.01, real=np.float64)
    assert st["ndt_iterations"] == ref["iterations"]
    da, dtv = sm.se3_error(R, ref["result"])
    assert da < 1e-4 and dtv < 1e-3, (da, dtv)
    assert abs(st["ndt_score"] - ref["score"]) < 1e-3 * ref["score"]
    # ---- GICP covariances of the filtered source and of a slab of the filtered target
    m.set_gicp_options(gicp_corr_dist_threshold=5.0, gicp_max_iterations=1)
    m.align(G)
    for which, cloud in ((0, os_), (1, ot)):
        got = m.get_covariances(which, len(cloud))
        pick = np.arange(0, len(cloud), max(1, len(cloud) // 40000))
        d, nn = cKDTree(cloud.astype(np.float64)).query(cloud[pick].astype(np.float64), k=21)
        P = cloud[nn[:, :20]]
        mean = P.astype(np.float64).sum(axis=1) / 20
        cov = (P[:, :, :, None] * P[:, :, None, :]).astype(np.float64).sum(axis=1) / 20 - mean[:, :, None] * mean[:, None, :]
        cov = 0.5 * (cov + np.transpose(cov, (0, 2, 1)))
        w, U = np.linalg.eigh(cov)
        col = np.argmin(np.abs(w), axis=1)
        u3 = np.take_along_axis(U, col[:, None, None], axis=2)[:, :, 0]
        want = np.eye(3)[None] - (1.0 - 1e-3) * u3[:, :, None] * u3[:, None, :]
        clear = (d[:, 20] - d[:, 19]) > 1e-4
        # the smallest-|eigenvalue| direction is itself ill-defined when two eigenvalues nearly tie: compare where it is not
        aw = np.sort(np.abs(w), axis=1)
        clear &= (aw[:, 1] - aw[:, 0]) > 1e-3 * aw[:, 2]
        err = np.abs(got[pick] - want).max(axis=(1, 2))
        assert clear.mean() > 0.6
        assert np.mean(err[clear] < 1e-5) > 0.99, (which, np.mean(err[clear] < 1e-5), np.sort(err[clear])[-5:])
    m.set_gicp_options(gicp_max_iterations=35)
    ok, R = m.align(G)
    da, dtv = sm.se3_error(R, T)
    print(f"config #5: {m.last_gicp_stats}, whole run vs truth {da:.2e} rad {dtv:.2e} m")
    assert ok and da < 5e-3 and dtv < 0.1
    # ---- the WHOLE matcher (filter -> NDT -> GICP with its BFGS line searches, ndt_gicp.cc:55-112) against the oracle at
    # north_star's tolerance.  A whole GICP run is in general only repeatable to GICP's own accuracy (the functor's float
    # transform decides Wolfe tests at its noise floor: tests/test_oracle_ndt_gicp.py, tests/test_ndt_gicp_gpu.py), but on this
    # case -- the BASELINE config the bench reports -- device and oracle take the same decisions throughout, so the bound is the
    # north star's: 1e-4 rad / 1e-3 m (measured 2.5e-6 rad / 6e-5 m), with equal cloud sizes, NDT iterations and GICP iterations and
    # comparable numbers of evaluated line-search states.
    st = dict(m.last_gicp_stats)
    tr = []
    g_o = ong.gicp_align(os_, ot, ref["result"].astype(np.float32), trace=tr)
    da, dtv = sm.se3_error(R, g_o["result"].astype(np.float64))
    # (the device evaluates f and the gradient in one launch per trial state; pcl asks for f and, at the same state, maybe for df:
    # what must agree is the number of STATES the two line searches visited)
    evals_o = int(sum(t_["points"] for t_ in tr))
    print(f"config #5 whole run vs oracle: {da:.2e} rad {dtv:.2e} m; GICP iterations {st['gicp_iterations']} / {g_o['iterations']}, "
          f"functor evaluations {st['gicp_function_evaluations']} / {evals_o}")
    assert da < 1e-4 and dtv < 1e-3, (da, dtv)
    assert st["n_source"] == len(os_) and st["n_target"] == len(ot)
    assert st["ndt_iterations"] == ref["iterations"] and st["gicp_iterations"] == g_o["iterations"]
    # (pcl asks its functor for f and for df separately and caches each by step length; the device answers both from one launch per
    # state: the two counts are of the same line searches but not of the same events -- 40 against 45 here)
    assert abs(st["gicp_function_evaluations"] - evals_o) <= max(5, evals_o // 5), (st["gicp_function_evaluations"], evals_o)
    assert abs(m.get_fitness_score() - float(np.exp(-g_o["score"]))) < 1e-4
    m.close()


@pytest.mark.parametrize("variant", ["faithful", "stress"])
def test_config2_icp_fast_both_target_variants(cfg2, variant):
    """SURVEY §8(d) cfg 2: target = scan A through CalculateNormals (N_t' ~ 21.6 k) [faithful] and scan A with a normal per
    point (N_t = 120 k) [stress]; 20 fixed iterations and the early-exit run, GPU vs the C oracle."""
    from oracle import cref
    c = cfg2
    if variant == "faithful":
        q, n = c["q"], c["n"]
    else:                                                   # every raw point keeps the normal of its kd leaf
        from scipy.spatial import cKDTree
        q = c["tgt"][:, :3].astype(np.float64)
        _, j = cKDTree(c["q"]).query(q)
        n = c["n"][j]
    src = c["src"][:, :3].astype(np.float64)
    m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q))
    m.set_input_source(c["src"]); m.set_input_target(q, n)
    for ee, mi in ((0, 20), (1, 100)):
        m.set_options(max_iteration=mi, early_exit=ee)
        ok, R = m.align(c["guess"])
        ref = cref.icp_fast_align(src, q, n, guess=c["guess"], max_iteration=mi, early_exit=bool(ee), nthreads=cref.usable_cores())
        assert m.last_stats[0]["iterations"] == ref["iterations"]
        da, dt = sm.se3_error(R, ref["result"])
        assert da < 1e-4 and dt < 1e-3, (variant, ee, da, dt)
        assert abs(m.get_fitness_score() - ref["score"]) < 1e-5
    m.close()


def test_ndt_gicp_eight_seeds_stages_asserted_whole_run_reported(capsys):
    """NdtWithGicp over EIGHT different scan / submap pairs (120k-pt scan vs a 500k-pt submap of five scans each: the matcher's
    stages at the size the front end hands it).  One seed says little about a matcher whose last stage -- GICP's BFGS -- is chaotic
    under legal float roundings (tests/test_oracle_ndt_gicp.py: the oracle's own fma / no-fma builds end 0.14 m apart on one
    pair).  So the deterministic stages are ASSERTED on every seed -- both down-sampled clouds bit-equal to PCL's serial
    ApproximateVoxelGrid, the stock-NDT stage's iteration count, pose (1e-4 rad / 1e-3 m) and fitness, the batch of eight returning
    the single calls' bits -- and the whole run is REPORTED: device vs the oracle next to the oracle's own fma vs no-fma spread on
    the same pair (the band any two builds of the reference agree in)."""
    from oracle import ndt_gicp as ong
    from oracle import ndt as ondt
    seeds = [11, 12, 13, 14, 15, 16, 17, 18]
    cases = [_submap(5, 500_000, sd) for sd in seeds]
    ns = max(len(c[0]) for c in cases); nt = max(len(c[1]) for c in cases)
    m = sm.NdtGicpHip(max_source_points=ns, max_target_points=nt, jobs=len(seeds))
    single = []
    rows = []
    for k, (src, tgt, T, G) in enumerate(cases):
        m.set_input_source(src, slot=k); m.set_input_target(tgt, slot=k)
    m1 = sm.NdtGicpHip(max_source_points=ns, max_target_points=nt)
    for k, (src, tgt, T, G) in enumerate(cases):
        m1.set_input_source(src); m1.set_input_target(tgt)
        # NDT stage alone (no correspondence inside the gate: GICP hands NDT's pose back)
        m1.set_gicp_options(gicp_corr_dist_threshold=1e-9)
        ok, R_ndt = m1.align(G)
        st = dict(m1.last_gicp_stats)
        ds, dt_ = m1.get_downsampled(0), m1.get_downsampled(1)
        os_ = ong.approximate_voxel_grid(src, 0.2)
        ot = ong.approximate_voxel_grid(tgt, 0.2)
        assert np.array_equal(ds, os_) and np.array_equal(dt_, ot), k
        ref = ondt.ndt_align(os_, ot, guess=G, trans_eps=0.01, real=np.float64)
        assert st["ndt_iterations"] == ref["iterations"], (k, st["ndt_iterations"], ref["iterations"])
        da, dtv = sm.se3_error(R_ndt, ref["result"])
        assert da < 1e-4 and dtv < 1e-3, (k, da, dtv)
        assert abs(st["ndt_score"] - ref["score"]) < 1e-3 * ref["score"], k
        # the whole matcher
        m1.set_gicp_options(gicp_corr_dist_threshold=5.0)
        ok, R = m1.align(G)
        single.append(R)
        runs = {mode: ong.gicp_align(os_, ot, ref["result"].astype(np.float32), transform_mode=mode) for mode in ("nofma", "fma")}
        d_dev = sm.se3_error(R, runs["fma"]["result"].astype(np.float64))
        d_own = sm.se3_error(runs["fma"]["result"].astype(np.float64), runs["nofma"]["result"].astype(np.float64))
        d_truth = sm.se3_error(R, T)
        rows.append((seeds[k], st["ndt_iterations"], d_dev, d_own, d_truth))
        assert ok and d_truth[0] < 5e-3 and d_truth[1] < 0.1, (k, d_truth)
    m1.close()
    Rb, sc, stb = m.align_batch(len(seeds), [c[3] for c in cases])
    for k in range(len(seeds)):
        assert Rb[k].tobytes() == single[k].tobytes(), k
    m.close()
    with capsys.disabled():
        print("\n[NdtWithGicp, 8 seeds] seed: NDT iterations | whole run device vs oracle (fma) | oracle fma vs no-fma | device vs truth")
        for sd, it, a, b, c in rows:
            print(f"  {sd}: {it} | {a[0]:.2e} rad {a[1]:.2e} m | {b[0]:.2e} rad {b[1]:.2e} m | {c[0]:.2e} rad {c[1]:.2e} m")
        inside = sum(1 for _, _, a, b, _ in rows if a[1] <= max(1e-3, 2.0 * b[1]) and a[0] <= max(1e-4, 2.0 * b[0]))
        print(f"  device within max(north-star tolerance, twice the oracle's own spread) of the oracle on {inside} of {len(rows)} seeds")
