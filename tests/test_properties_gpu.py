"""Property tests (hypothesis) of the bit-exact device paths against their oracles on random inputs: filter chains,
the ApproximateVoxelGrid restatement, and the exact-NN variants.  Few examples each: every example is a GPU round trip."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

import staticmapping_amd as sm
from staticmapping_amd import filters as df
from oracle import filters as of, ndt_gicp as ong

pytestmark = pytest.mark.gpu
import os
_N = int(os.environ.get("SMHIP_HYP_EXAMPLES", "20"))
_SETTINGS = dict(max_examples=_N, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def handle():
    m = sm.NdtGicpHip(max_source_points=32768, max_target_points=32768)
    yield m
    m.close()


def _cloud(rng, n, scale, specials):
    c = np.zeros((n, 5), np.float32)
    c[:, :3] = rng.normal(0.0, scale, (n, 3))
    c[:, 3] = rng.uniform(0, 255, n)
    c[:, 4] = rng.uniform(0, 1, n)
    if specials and n >= 8:
        k = rng.integers(0, n, 4)
        c[k[0], 0] = np.nan; c[k[1], 1] = np.inf; c[k[2], 2] = -np.inf; c[k[3], :3] = 0.0
    return c


_filter = st.one_of(
    st.tuples(st.just("Range"), st.floats(0, 30), st.floats(31, 200)).map(lambda t: ("Range", dict(min_range=t[1], max_range=t[2]))),
    st.tuples(st.integers(0, 2), st.floats(-50, 0), st.floats(1, 60)).map(lambda t: ("AxisRange", dict(axis_index=t[0], min=t[1], max=t[2]))),
    st.tuples(st.floats(-40, 0), st.floats(1, 40)).map(lambda t: ("BoundingBoxRemoval", dict(min_x=t[0], max_x=t[1], min_y=t[0] / 2, max_y=t[1] / 2, min_z=-5.0, max_z=5.0))),
    st.tuples(st.floats(0.05, 0.95), st.integers(0, 2**31 - 1)).map(lambda t: ("RandomSampler", dict(sampling_rate=t[0], seed=t[1]))),
)


def _oracle_filter(name, kw):
    t = df.NAMES[name]
    f = of.default(t)
    f.update(kw)
    return f


@settings(**_SETTINGS)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(0, 20000), scale=st.sampled_from([0.3, 5.0, 40.0]), specials=st.booleans(),
       chain=st.lists(_filter, min_size=0, max_size=4), voxel=st.one_of(st.none(), st.sampled_from([0.1, 0.37, 2.5])))
def test_filter_chains_are_bit_exact(handle, seed, n, scale, specials, chain, voxel):
    rng = np.random.default_rng(seed)
    raw = _cloud(rng, n, scale, specials and voxel is None)          # lround of a non-finite value is undefined in the reference
    dchain = [df.make_filter(name, **kw) for name, kw in chain]
    ochain = [_oracle_filter(name, kw) for name, kw in chain]
    if voxel is not None:
        dchain.append(df.make_filter("VoxelGrid", voxel_size=voxel)); ochain.append(dict(of.default(of.VOXEL_GRID), voxel_size=voxel))
    got, gsrc = df.run_chain(handle, raw, dchain)
    want, wsrc = of.run_chain(raw, ochain)
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(gsrc, wsrc)


@settings(**_SETTINGS)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(30, 20000), scale=st.sampled_from([0.15, 1.0, 25.0]), leaf=st.sampled_from([0.2, 0.5]))
def test_approximate_voxel_grid_is_bit_exact(handle, seed, n, scale, leaf):
    """pcl::ApproximateVoxelGrid restated in parallel: same centroids, same order, for clouds that revisit voxels
    (small scale: few voxels, many evictions) and for sparse ones."""
    rng = np.random.default_rng(seed)
    pts = rng.normal(0.0, scale, (n, 3)).astype(np.float32)
    handle.set_gicp_options(voxel_resolution=leaf, use_ndt=0, gicp_max_iterations=1)
    handle.set_input_source(pts); handle.set_input_target(pts)
    try:
        handle.align()
    except sm.SmhipError:
        pass                                       # fewer than k points after the filter: GICP refuses, the filter ran
    want = ong.approximate_voxel_grid(pts, leaf)
    assert np.array_equal(handle.get_downsampled(1), want)
    assert np.array_equal(handle.get_downsampled(0), want)
    handle.set_gicp_options(voxel_resolution=0.2, use_ndt=1, gicp_max_iterations=35)


@settings(max_examples=max(8, _N // 3), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), ns=st.integers(1, 6000), nt=st.integers(1, 6000), spread=st.sampled_from([0.5, 8.0, 60.0]),
       offset=st.sampled_from([0.0, 0.3, 25.0]))
def test_every_nn_variant_agrees_with_brute_force(seed, ns, nt, spread, offset):
    """Exact 1-NN with the smallest-position tie rule: ball / LDS-table / ring / cooperative ring / fallback / brute force
    return the same ids and bit-identical distances for random clouds, including sources far outside the target's box."""
    rng = np.random.default_rng(seed)
    tgt = rng.normal(0.0, spread, (nt, 3))
    src = rng.normal(0.0, spread, (ns, 3)) + offset
    nrm = np.tile([0.0, 0.0, 1.0], (nt, 1))
    ref = None
    for opts in (dict(nn_mode=0), dict(), dict(no_lds_table=1), dict(use_ball=0), dict(grid_cell=1.0), dict(grid_max_ring=1)):
        m = sm.IcpFastHip(max_source_points=ns, max_target_points=nt, **opts)
        m.set_input_source(src); m.set_input_target(tgt, nrm)
        ids, d2 = m.find_closests(np.eye(4), ns)
        m.close()
        if ref is None:
            ref = (ids, d2)
        assert np.array_equal(ids, ref[0]), opts
        assert np.array_equal(d2.view(np.uint32), ref[1].view(np.uint32)), opts


@settings(max_examples=max(6, _N // 4), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(800, 6000), yaw=st.floats(-3.0, 3.0), tx=st.floats(-0.3, 0.3), rho=st.sampled_from([0.5, 0.7, 0.9]))
def test_align_is_identical_across_search_variants(seed, n, yaw, tx, rho):
    """Whatever way the matches are found (fused ball search, certificates off, global-memory variant, the two-launch
    converged path from iteration 1, plain ring search, brute force), a whole Align returns the same bits: the matches that
    survive the trimming are the same with the same distances, and every sum runs in a fixed order."""
    from staticmapping_amd import synth
    tgt, src, T = synth.three_planes_pair(n, seed=seed % 1000, sigma=0.01)
    q, nr = sm.calculate_normals(tgt[:, :3].astype(np.float64))
    guess = synth.make_pose(t=(tx, 0.05, 0.0), rpy_deg=(0, 0, yaw))
    ref = None
    for opts in (dict(), dict(no_certify=1), dict(no_lds_table=1), dict(split_after=1), dict(split_after=3, no_lds_table=1),
                 dict(use_ball=0), dict(nn_mode=0), dict(ball_radius=0.05)):
        # (as separate launches per iteration: the search variants share every sum's order there)
        m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q), max_iteration=25, early_exit=1,
                          dist_outlier_ratio=rho, no_single_kernel=1, **opts)
        m.set_input_source(src); m.set_input_target(q, nr)
        ok, R = m.align(guess)
        st_ = m.last_stats[0]
        m.close()
        key = (R.tobytes(), st_["iterations"], st_["kept"], st_["limit_d2"])
        if ref is None:
            ref = key
            R_ref = R
        assert key == ref, (opts, st_)
    # the same Align as ONE cooperative launch (csrc/icp_one.hip): the same matches and kept sets, the sums in another fixed order
    m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q), max_iteration=25, early_exit=1, dist_outlier_ratio=rho)
    m.set_input_source(src); m.set_input_target(q, nr)
    ok, R = m.align(guess)
    st_ = m.last_stats[0]
    ok2, R2 = m.align(guess)
    m.close()
    assert R.tobytes() == R2.tobytes()
    assert st_["iterations"] == ref[1] and abs(st_["kept"] - ref[2]) <= 2, (st_, ref[1:])
    da, dt = sm.se3_error(R, R_ref)
    assert da < 1e-10 and dt < 1e-10, (da, dt)


@settings(max_examples=max(6, _N // 4), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(800, 6000), yaw=st.floats(-30.0, 30.0), tx=st.floats(-1.0, 1.0),
       rho=st.sampled_from([0.5, 0.9, 1.0]), far=st.sampled_from([0.0, 1000.0]))
def test_certificates_change_no_bit_far_from_the_origin_and_from_poor_guesses(seed, n, yaw, tx, rho, far):
    """The same bit equality where the certificates have least room: clouds a kilometre from the origin (float32 inputs there
    are 6e-5 m apart; the device works on the centred clouds), guesses up to 30 degrees and a metre off (the pose moves far
    in every iteration, most certificates fail and many queries fall outside every search ball), and trimming ratios up to
    1.0 (nothing trimmed: every lower-bounded match has to be refined)."""
    from staticmapping_amd import synth
    tgt, src, T = synth.three_planes_pair(n, seed=seed % 1000, sigma=0.01)
    off = np.array([far, -0.7 * far, 0.03 * far])
    q, nr = sm.calculate_normals(tgt[:, :3].astype(np.float64) + off)
    src = src.copy()
    src[:, :3] = (src[:, :3].astype(np.float64) + off).astype(np.float32)
    # the guess turns about the clouds' own position, not about the far-away origin
    P = synth.make_pose(t=(tx, 0.05, 0.0), rpy_deg=(0, 0, yaw))
    C = np.eye(4); C[:3, 3] = off
    Ci = np.eye(4); Ci[:3, 3] = -off
    guess = C @ P @ Ci
    ref = None
    for opts in (dict(), dict(no_certify=1), dict(split_after=1), dict(no_lds_table=1), dict(use_ball=0)):
        m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q), max_iteration=25, early_exit=1,
                          dist_outlier_ratio=rho, no_single_kernel=1, **opts)
        m.set_input_source(src); m.set_input_target(q, nr)
        ok, R = m.align(guess)
        st_ = m.last_stats[0]
        ids, d2 = m.get_matches(len(src))
        m.close()
        key = (R.tobytes(), st_["iterations"], st_["kept"], st_["limit_d2"], st_["status"])
        if ref is None:
            ref = key
            R_ref = R
        assert key == ref, (opts, st_)
    # ... and as ONE cooperative launch (the bounds refined by the queries' owners there): same kept sets, sums in another order
    m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q), max_iteration=25, early_exit=1, dist_outlier_ratio=rho)
    m.set_input_source(src); m.set_input_target(q, nr)
    ok, R = m.align(guess)
    st_ = m.last_stats[0]
    m.close()
    assert st_["iterations"] == ref[1] and abs(st_["kept"] - ref[2]) <= 2 and st_["status"] == ref[4], (st_, ref[1:])
    da, dt = sm.se3_error(R, R_ref)
    assert da < 1e-9 and dt < 1e-9, (da, dt)
