"""registrators::Ndt parity: voxel table, one computeDerivatives evaluation, and the full Align,
HIP path (through the C ABI) vs the numpy restatement in oracle/ndt.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ndt_case():
    """Source scan vs a target merged from 3 scans (a small submap), guess 0.3 m short of the truth."""
    from staticmapping_amd import synth
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.5 * k)) for k in range(4)]
    scans = [synth.velodyne_scan(scene, P, seed=10 + k, n_points=30000) for k, P in enumerate(poses)]
    tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:3], poses[:3])])
    tgt = np.concatenate([tgt, np.zeros((len(tgt), 1))], axis=1).astype(np.float32)
    T = poses[3]
    G = T.copy(); G[0, 3] -= 0.3
    return dict(src=scans[3], tgt=tgt, T=T, guess=G)


@pytest.fixture(scope="module")
def ndt_matcher(ndt_case):
    import staticmapping_amd as sm
    m = sm.NdtHip(max_source_points=len(ndt_case["src"]), max_target_points=len(ndt_case["tgt"]))
    m.set_input_source(ndt_case["src"])
    m.set_input_target(ndt_case["tgt"])
    yield m
    m.close()


@pytest.fixture(scope="module")
def oracle_grid(ndt_case):
    from oracle import ndt as ondt
    return ondt.VoxelGrid(ndt_case["tgt"])


def test_voxel_table_matches_applyfilter(ndt_matcher, oracle_grid):
    n = ndt_matcher.build_voxels()
    keys, counts, means, icov, cent = ndt_matcher.get_voxels(n)
    searchable = (counts >= 6) | (counts == -1)
    got = dict(zip(keys[searchable].tolist(), np.nonzero(searchable)[0].tolist()))
    assert sorted(got) == sorted(oracle_grid.key.tolist())           # same voxels with >= 6 points
    idx = np.array([got[k] for k in oracle_grid.key.tolist()])
    assert np.allclose(means[idx], oracle_grid.mean, rtol=0, atol=1e-9)
    assert np.allclose(cent[idx], oracle_grid.centroid, rtol=0, atol=2e-4)
    ic = oracle_grid.icov
    ref6 = np.stack([ic[:, 0, 0], ic[:, 0, 1], ic[:, 0, 2], ic[:, 1, 1], ic[:, 1, 2], ic[:, 2, 2]], axis=1)
    ok = oracle_grid.valid
    assert ((counts[idx] == -1) == ~ok).all()
    scale = np.abs(ref6[ok]).max(axis=1, keepdims=True)
    assert (np.abs(icov[idx][ok] - ref6[ok]) <= 2e-5 * scale + 1e-6).all()


def test_compute_derivatives_matches_oracle(ndt_matcher, oracle_grid, ndt_case):
    from oracle import ndt as ondt
    ndt_matcher.build_voxels()
    d1, d2, _ = ondt.gauss_constants()
    for pose in ([2.1, 0.05, 0.0, 0.0, 0.0, 0.015], [2.3, -0.1, 0.02, 0.004, -0.003, 0.03]):
        p = np.array(pose)
        T = ondt.pose_to_matrix_f32(p)
        trans = ondt.transform_cloud_f32(ndt_case["src"], T)
        s_o, g_o, H_o, _ = ondt.compute_derivatives(oracle_grid, ndt_case["src"], trans, p, d1, d2, True)
        s_g, g_g, H_g = ndt_matcher.compute_derivatives(p, True)
        assert abs(s_g - s_o) <= 1e-4 * abs(s_o)
        assert np.allclose(g_g, g_o, rtol=2e-3, atol=1e-3 * np.abs(g_o).max())
        assert np.allclose(H_g, H_o, rtol=2e-3, atol=1e-3 * np.abs(H_o).max())


def test_ndt_align_parity(ndt_matcher, oracle_grid, ndt_case):
    import staticmapping_amd as sm
    from oracle import ndt as ondt
    ok, R = ndt_matcher.align(ndt_case["guess"])
    ref = ondt.ndt_align(ndt_case["src"], ndt_case["tgt"], guess=ndt_case["guess"], grid=oracle_grid)
    st = ndt_matcher.last_ndt_stats
    assert st["iterations"] == ref["iterations"] and st["derivative_calls"] == ref["derivative_calls"]
    da, dt = sm.se3_error(R, ref["result"])
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(ndt_matcher.get_fitness_score() - ref["score"]) <= 1e-3 * ref["score"]
    assert abs(st["trans_probability"] - ref["trans_probability"]) <= 1e-4 * abs(ref["trans_probability"])
    # NDT's clamped step (0.05..0.1) leaves a few cm by construction (oracle shows the same)
    da, dt = sm.se3_error(R, ndt_case["T"])
    assert da < 2e-3 and dt < 0.1


def test_ndt_identity_guess_and_errors(ndt_case):
    import staticmapping_amd as sm
    from oracle import ndt as ondt
    m = sm.NdtHip(max_source_points=len(ndt_case["src"]), max_target_points=len(ndt_case["tgt"]))
    with pytest.raises(sm.SmhipError) as e:          # Align before SetInput* -> false in the reference (ndt.cc:40-42)
        m.align()
    assert e.value.status == 4
    sub = ndt_case["src"][::3]
    m.set_input_source(sub); m.set_input_target(ndt_case["tgt"])
    G = ndt_case["T"].copy()                          # start at the truth: Newton step < 0.05 -> min step kicks in
    ok, R = m.align(G)
    ref = ondt.ndt_align(sub, ndt_case["tgt"], guess=G)
    assert m.last_ndt_stats["iterations"] == ref["iterations"]
    da, dt = sm.se3_error(R, ref["result"])
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    m.close()


def test_lockstep_batch_equals_the_single_calls_bit_for_bit(ndt_case):
    """smhip_ndt_align_batch: K Aligns advance in lock-step (each round's computeDerivatives calls of all running pairs are one
    launch), every pair through exactly the evaluation sequence of its own Newton / More-Thuente state machine
    (pclomp/ndt_omp_impl.hpp:81-171, 757-916).  Five different pairs -- different sources, targets, guesses, so different
    iteration counts: the pairs drop out of the rounds at different times -- against the same five as single calls: poses,
    fitness scores and statistics bit for bit; also on a slot range that does not start at 0, and with the tables kept."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth
    src, tgt, T = ndt_case["src"], ndt_case["tgt"], ndt_case["T"]
    cases = []
    for k in range(5):
        s = src[k::(1 + k % 3)]                                              # different sizes
        t = tgt[(k % 2)::(1 + k % 2)]
        g = T @ synth.make_pose(t=(-0.1 - 0.07 * k, 0.03 * k, 0.0), rpy_deg=(0, 0, 0.2 * k))
        cases.append((s, t, g))
    single = []
    m1 = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
    for s, t, g in cases:
        m1.set_input_source(s); m1.set_input_target(t)
        ok, R = m1.align(g)
        single.append((R, m1.get_fitness_score(), dict(m1.last_ndt_stats)))
    m1.close()
    assert len({st["iterations"] for _, _, st in single}) > 1                # the state machines do not finish together
    mb = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt), pair_slots=7)
    for first in (0, 2):
        for k, (s, t, g) in enumerate(cases):
            mb.set_input_source(s, slot=first + k); mb.set_input_target(t, slot=first + k)
        for rep in range(2):                                                 # second pass: every table is current and kept
            R, sc, st = mb.align_batch(5, [c[2] for c in cases], first_slot=first)
            for k in range(5):
                assert R[k].tobytes() == single[k][0].tobytes(), (first, rep, k)
                assert sc[k] == single[k][1] and st[k] == single[k][2], (first, rep, k, st[k], single[k][2])
    mb.close()


def test_fitness_score_is_the_exact_nearest_neighbour_mean(ndt_case):
    """pcl::Registration::getFitnessScore (ndt.cc:60) through the table's own search structure (ndt_fit_near / ndt_fit_mid / ndt_fit_far):
    for poses that leave the scan on the submap, half off it, 40 m away from it (every query goes through the far pass) and turned by
    30 degrees, the reported score against brute force -- a k-d tree over the raw target, the source moved by the float32 of the pose
    the Align returned -- to float rounding of the single distances."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth
    from scipy.spatial import cKDTree
    src, tgt, T = ndt_case["src"], ndt_case["tgt"], ndt_case["T"]
    tree = cKDTree(tgt[:, :3].astype(np.float64))
    m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
    m.set_input_source(src); m.set_input_target(tgt)
    guesses = [T, T @ synth.make_pose(t=(6.0, -3.0, 0.2)), T @ synth.make_pose(t=(40.0, 25.0, 1.0)), T @ synth.make_pose(rpy_deg=(0, 0, 30.0)),
               synth.make_pose(t=(500.0, 0.0, 0.0))]
    for k, G in enumerate(guesses):
        for cache in (False, True):
            m.set_target_cache(cache)
            ok, R = m.align(G)
            Rf = R.astype(np.float32)
            moved = (src[:, :3].astype(np.float32) @ Rf[:3, :3].T + Rf[:3, 3]).astype(np.float64)
            d, _ = tree.query(moved)
            want = float(np.mean(d * d))
            got = m.get_fitness_score()
            assert abs(got - want) <= 2e-5 * want + 1e-9, (k, cache, got, want)
    m.close()
