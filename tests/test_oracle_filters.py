"""The pre-filter oracle (oracle/filters.py) replays the known answers of the reference's own filter tests
(/root/reference/pre_processors/test/test_filter_*.cc) -- this is what pins it.  The reference draws its random test
clouds from an unseeded generator (test/test_helper.cc:30-61: uniform [0, 100) per coordinate), so the clouds here
follow the same law with a fixed seed; every assertion below is one the reference test makes."""
import numpy as np
import pytest

from oracle import filters as of


def random_inner_cloud(n, seed=0):
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 5), np.float32)
    c[:, :4] = rng.uniform(0.0, 100.0, (n, 4))
    return c


def test_range_known_answers():                       # test_filter_range.cc:37-73
    raw = random_inner_cloud(1000)
    out, src = of.run_chain(raw, [of.default(of.RANGE)])
    assert np.array_equal(out, raw) and np.array_equal(src, np.arange(1000))
    f = of.default(of.RANGE); f.update(min_range=20.0, max_range=80.0)
    out, _ = of.run_chain(raw, [f])
    r = np.sqrt((out[:, :3].astype(np.float32) ** 2).sum(axis=1))
    assert len(out) > 0 and np.all(r <= 80.0) and np.all(r >= 20.0)


def test_axis_range_known_answers():                  # test_filter_axis_range.cc:36-145
    bad = of.default(of.AXIS_RANGE); bad.update(min=90.0, max=80.0)
    assert not of.config_valid(bad)
    bad = of.default(of.AXIS_RANGE); bad.update(min=10.0, max=80.0, axis_index=-1)
    assert not of.config_valid(bad)
    raw = random_inner_cloud(1000, 1)
    out, _ = of.run_chain(raw, [of.default(of.AXIS_RANGE)])
    assert np.array_equal(out, raw)
    f = of.default(of.AXIS_RANGE); f.update(min=50.0)
    out, _ = of.run_chain(raw, [f]); assert np.all(out[:, 2] >= 50.0)
    f.update(max=80.0)
    out, _ = of.run_chain(raw, [f]); assert np.all((out[:, 2] >= 50.0) & (out[:, 2] <= 80.0))
    f.update(min=60.0, axis_index=1)
    out, _ = of.run_chain(raw, [f]); assert np.all((out[:, 1] >= 60.0) & (out[:, 1] <= 80.0))
    f.update(min=10.0, max=70.0, axis_index=0)
    out, _ = of.run_chain(raw, [f]); assert np.all((out[:, 0] >= 10.0) & (out[:, 0] <= 70.0))


def test_bounding_box_known_answers():                # test_filter_bounding_box.cc:38-118
    for kw in (dict(min_x=90.0, max_x=80.0), dict(min_x=10.0, max_x=80.0, min_y=10.0, max_y=0.0), dict(min_z=90.0, max_z=80.0)):
        f = of.default(of.BOUNDING_BOX_REMOVAL); f.update(kw)
        assert not of.config_valid(f)
    raw = random_inner_cloud(1000, 2)
    out, _ = of.run_chain(raw, [of.default(of.BOUNDING_BOX_REMOVAL)])
    assert len(out) == 0                                                       # the default box contains everything
    f = of.default(of.BOUNDING_BOX_REMOVAL); f.update(min_x=0.0, max_x=100.0, min_y=0.0, max_y=100.0, min_z=0.0, max_z=100.0)
    out, _ = of.run_chain(raw, [f]); assert len(out) == 0
    f.update(min_x=10.0, max_x=80.0, min_y=20.0, max_y=70.0, min_z=30.0, max_z=80.0)
    out, _ = of.run_chain(raw, [f])
    inside = np.all((out[:, :3] >= [10.0, 20.0, 30.0]) & (out[:, :3] <= [80.0, 70.0, 80.0]), axis=1)
    assert len(out) > 0 and not inside.any()


def test_random_sampler_known_answers():              # test_filter_random_sample.cc:36-86
    assert of.config_valid(of.default(of.RANDOM_SAMPLER))
    f = of.default(of.RANDOM_SAMPLER); f.update(sampling_rate=1.5)
    assert not of.config_valid(f)
    raw = random_inner_cloud(100000, 3)
    for seed in range(100):
        f = of.default(of.RANDOM_SAMPLER); f.update(sampling_rate=0.5, seed=seed)
        out, src = of.run_chain(raw, [f])
        assert 0.48 < len(out) / 100000 < 0.52
        assert np.array_equal(out, raw[src]) and np.all(np.diff(src) > 0)


def test_voxel_grid_known_answers():                  # test_filter_voxel_grid.cc:35-100
    f = of.default(of.VOXEL_GRID); f.update(voxel_size=0.0)
    assert not of.config_valid(f)
    f.update(voxel_size=10.0)
    assert of.config_valid(f)
    raw = np.zeros((100, 5), np.float32)
    k = 0
    for x in range(10):
        for y in range(10):
            raw[k, :3] = (np.float32(x) * np.float32(0.1) + np.float32(0.02), np.float32(y) * np.float32(0.1) + np.float32(0.02), 0.1)
            k += 1
    for size, want in ((0.1, 100), (0.2, 36), (0.4, 9)):
        f = of.default(of.VOXEL_GRID); f.update(voxel_size=size)
        out, src = of.run_chain(raw, [f])
        assert len(out) == want, (size, len(out))
        assert np.all(src == -1)


def test_chain_of_the_kitti_config():                 # config/lidar_only_kitti.xml:18-41 through Factory::Filter
    raw = random_inner_cloud(20000, 4); raw[:, :3] -= 50.0
    r, a, s = of.default(of.RANGE), of.default(of.AXIS_RANGE), of.default(of.RANDOM_SAMPLER)
    r.update(min_range=5.0); a.update(min=-2.0); s.update(sampling_rate=0.5, seed=7)
    out, src = of.run_chain(raw, [r, a, s])
    assert np.array_equal(out, raw[src])
    norm = np.sqrt((out[:, :3] ** 2).sum(axis=1))
    assert np.all(norm >= 5.0) and np.all(out[:, 2] >= -2.0)
    step1, _ = of.run_chain(raw, [r, a])
    assert 0.45 < len(out) / len(step1) < 0.55
    bad = of.default(of.AXIS_RANGE); bad.update(min=1.0, max=0.0)
    with pytest.raises(ValueError):
        of.run_chain(raw, [bad])
