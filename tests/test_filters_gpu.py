"""Device pre-filters (smhip_filter_*) vs the pinned oracle (oracle/filters.py): bit-exact clouds and indices, the
reference tests' known answers through the C ABI, the chain of config/lidar_only_kitti.xml, and the device-resident
hand-over of the filtered cloud to the matcher."""
import numpy as np
import pytest

import staticmapping_amd as sm
from staticmapping_amd import filters as df, synth
from oracle import filters as of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    m = sm.IcpFastHip(pair_slots=2, max_source_points=131072, max_target_points=131072)
    yield m
    m.close()


def cloud(n, seed=0, lo=0.0, hi=100.0):
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 5), np.float32)
    c[:, :4] = rng.uniform(lo, hi, (n, 4))
    c[:, 4] = (np.arange(n) / n).astype(np.float32)
    return c


def oracle_desc(d):
    t = d.type
    f = of.default(t)
    names = {of.RANGE: ("min_range", "max_range"), of.AXIS_RANGE: ("min", "max"), of.RANDOM_SAMPLER: ("sampling_rate",),
             of.VOXEL_GRID: ("voxel_size",), of.BOUNDING_BOX_REMOVAL: ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")}[t]
    for k, nme in enumerate(names):
        f[nme] = d.p[k]
    if t == of.AXIS_RANGE: f["axis_index"] = d.axis_index
    if t == of.RANDOM_SAMPLER: f["seed"] = d.seed
    return f


def check_chain(matcher, raw, chain, ordered=True):
    got, gsrc = df.run_chain(matcher, raw, chain)
    want, wsrc = of.run_chain(raw if raw.shape[1] == 5 else of.with_factor(raw), [oracle_desc(d) for d in chain])
    assert got.shape == want.shape, (got.shape, want.shape)
    if ordered:
        assert np.array_equal(got, want, equal_nan=True) and np.array_equal(gsrc, wsrc)
    else:                                              # VoxelGrid: the reference's order is unspecified; both sort by voxel
        assert np.array_equal(got, want) and np.all(gsrc == -1)
    return got


def test_defaults_are_identity_and_config_validity(matcher):
    raw = cloud(1000)
    for name in ("Range", "AxisRange", "RandomSampler"):
        assert np.array_equal(check_chain(matcher, raw, [df.make_filter(name)]), raw)
    assert len(check_chain(matcher, raw, [df.make_filter("BoundingBoxRemoval")])) == 0      # test_filter_bounding_box.cc:73-77
    assert not df.config_valid(df.make_filter("AxisRange", min=90.0, max=80.0))
    assert not df.config_valid(df.make_filter("AxisRange", min=10.0, max=80.0, axis_index=-1))
    assert not df.config_valid(df.make_filter("RandomSampler", sampling_rate=1.5))
    assert not df.config_valid(df.make_filter("VoxelGrid", voxel_size=0.0))
    assert df.config_valid(df.make_filter("VoxelGrid", voxel_size=10.0))
    assert not df.config_valid(df.make_filter("BoundingBoxRemoval", min_z=90.0, max_z=80.0))
    with pytest.raises(sm.SmhipError):                  # an invalid filter never runs
        df.run_chain(matcher, raw, [df.make_filter("AxisRange", min=1.0, max=0.0)])
    with pytest.raises(KeyError):                       # unknown parameter name = CHECK failure in the reference
        df.make_filter("Range", min=1.0)


def test_each_filter_bit_exact(matcher):
    raw = cloud(50000, 1)
    check_chain(matcher, raw, [df.make_filter("Range", min_range=20.0, max_range=80.0)])
    for axis in (0, 1, 2):
        check_chain(matcher, raw, [df.make_filter("AxisRange", min=10.0 + axis, max=70.0, axis_index=axis)])
    check_chain(matcher, raw, [df.make_filter("BoundingBoxRemoval", min_x=10.0, max_x=80.0, min_y=20.0, max_y=70.0, min_z=30.0, max_z=80.0)])
    for seed in (0, 1, 12345):
        got = check_chain(matcher, raw, [df.make_filter("RandomSampler", sampling_rate=0.5, seed=seed)])
        assert 0.48 < len(got) / len(raw) < 0.52         # test_filter_random_sample.cc:78-84


def test_voxel_grid_known_answers_and_bits(matcher):
    raw = np.zeros((100, 5), np.float32)
    k = 0
    for x in range(10):
        for y in range(10):
            raw[k, :3] = (np.float32(x) * np.float32(0.1) + np.float32(0.02), np.float32(y) * np.float32(0.1) + np.float32(0.02), 0.1)
            k += 1
    for size, want in ((0.1, 100), (0.2, 36), (0.4, 9)):   # test_filter_voxel_grid.cc:62-99
        got = check_chain(matcher, raw, [df.make_filter("VoxelGrid", voxel_size=size)], ordered=False)
        assert len(got) == want
    big = cloud(60000, 2, -40.0, 40.0)
    check_chain(matcher, big, [df.make_filter("VoxelGrid", voxel_size=0.5)], ordered=False)
    check_chain(matcher, big, [df.make_filter("VoxelGrid", voxel_size=3.0)], ordered=False)


def test_kitti_config_chain_and_edges(matcher):
    a, b, T = synth.scan_pair("cfg2", n_points=40000)
    xml = """<filters><filter name="Range" ><param type="1" name="min_range"> 5. </param></filter>
             <filter name="AxisRange" ><param type="1" name="min"> -2. </param></filter>
             <filter name="GroundRemoval2" ><param type="1" name="r_min"> 0.1 </param></filter>
             <filter name="RandomSampler" ><param type="1" name="sampling_rate"> 0.5 </param></filter></filters>"""
    chain = df.chain_from_xml(xml, seed=3)
    assert [d.type for d in chain] == [df.RANGE, df.AXIS_RANGE, df.RANDOM_SAMPLER]          # unsupported names are skipped
    got = check_chain(matcher, np.ascontiguousarray(a[:, :4]), chain)                        # KITTI rows: factor = i / n
    assert 0.2 < len(got) / len(a) < 0.6
    # edge cases: empty input, everything dropped mid-chain, non-finite rows
    assert len(check_chain(matcher, np.zeros((0, 5), np.float32), chain)) == 0
    assert len(check_chain(matcher, cloud(1000, 5), [df.make_filter("Range", min_range=1e6), df.make_filter("VoxelGrid")])) == 0
    bad = cloud(2000, 6)
    bad[::7, 0] = np.nan; bad[3::11, 2] = np.inf
    check_chain(matcher, bad, [df.make_filter("Range", min_range=10.0, max_range=150.0), df.make_filter("AxisRange", min=5.0)])
    check_chain(matcher, bad, [df.make_filter("BoundingBoxRemoval", min_x=10.0, max_x=80.0, min_y=20.0, max_y=70.0, min_z=30.0, max_z=80.0)])


def test_filtered_cloud_feeds_the_matcher_without_a_host_hop(matcher):
    a, b, T = synth.scan_pair("cfg2", n_points=40000)
    chain = [df.make_filter("Range", min_range=5.0), df.make_filter("AxisRange", min=-2.0),
             df.make_filter("RandomSampler", sampling_rate=0.5, seed=9)]
    q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
    guess = synth.make_pose(t=(0.6, 0, 0))
    matcher.set_options(max_iteration=30, early_exit=1)
    matcher.set_input_target(q, n)
    filtered, _ = df.run_chain(matcher, np.ascontiguousarray(b[:, :4]), chain)
    df.output_to_source(matcher, 0)                                            # device-resident hand-over
    ok, R1 = matcher.align(guess)
    matcher.set_input_source(np.ascontiguousarray(filtered[:, :3]))            # the same cloud through the host
    ok, R2 = matcher.align(guess)
    assert np.array_equal(R1, R2)
    da, dt = sm.se3_error(R1, T)
    assert da < 2e-3 and dt < 0.05


def test_filter_error_conventions(matcher):
    import ctypes
    from staticmapping_amd import _capi
    raw = cloud(100)
    with pytest.raises(sm.SmhipError):                          # rows must be x y z intensity [factor]
        df.run_chain(matcher, raw[:, :3], [df.make_filter("Range")])
    with pytest.raises(sm.SmhipError):                          # larger than the handle's capacity
        df.run_chain(matcher, np.zeros((200000, 5), np.float32), [df.make_filter("Range")])
    d = _capi.FilterDesc(); d.type = 99
    assert not df.config_valid(d)
    with pytest.raises(sm.SmhipError):
        df.run_chain(matcher, raw, [d])
    df.run_chain(matcher, raw, [df.make_filter("Range", min_range=1e9)])          # everything dropped ...
    with pytest.raises(sm.SmhipError):                          # ... cannot become a matcher's source
        df.output_to_source(matcher, 0)
    got, src = df.run_chain(matcher, raw, [])                   # an empty chain is the identity (filter_factory.cc:90-93)
    assert np.array_equal(got, raw) and np.array_equal(src, np.arange(100))
