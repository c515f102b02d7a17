"""MultiResolutionVoxelMap (builder/multi_resolution_voxel_map.cc:36-170): the two restatements of the oracle -- plain C
(oracle/csrc/smref_mrvm.c, what the GPU tests check the device against) and plain Python (oracle/mrvm.py) -- against each
other and against cases small enough to work out by hand.  The reference has no test or fixture for this class."""
import numpy as np
import pytest

from oracle import cref, mrvm as pm


def _sorted_output(out):
    return out[np.lexsort((out[:, 3], out[:, 2], out[:, 1], out[:, 0]))] if len(out) else out


def test_bresenham_known_rays():
    # along one axis: every cell between the two ends, inclusive
    assert pm.bresenham((0.05, 0.05, 0.05), (0.45, 0.05, 0.05), 0.1) == [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0)]
    # negative coordinates floor away from zero (math.cc:42-44)
    assert pm.bresenham((-0.05, 0.0, 0.0), (-0.25, 0.0, 0.0), 0.1) == [(-1, 0, 0), (-2, 0, 0), (-3, 0, 0)]
    # a diagonal: dm = 4 steps, the minor axis advances when its error term runs out
    ray = pm.bresenham((0.0, 0.0, 0.0), (0.45, 0.25, 0.0), 0.1)
    assert ray[0] == (0, 0, 0) and ray[-1] == (4, 2, 0) and len(ray) == 5
    assert all(abs(b[0] - a[0]) <= 1 and abs(b[1] - a[1]) <= 1 for a, b in zip(ray, ray[1:]))
    # start and end in one cell
    assert pm.bresenham((0.01, 0.01, 0.01), (0.09, 0.09, 0.09), 0.1) == [(0, 0, 0)]


def test_probability_tables_by_hand():
    m = pm.Mrvm(hit_prob=0.55, miss_prob=0.48)
    # odds_table_[128] = log(0.5 / 0.5) = 0; a hit from "unknown": p = 1 - 1 / (1 + exp(log(0.55 / 0.45))) = 0.55 -> uint8(0.55 * 256) = 140
    assert float(m.odds[128]) == 0.0
    assert int(np.float32(m._update(128, True) * np.float32(256))) == 140
    # a miss from "unknown": 0.48 * 256 = 122.88 -> 122
    assert int(np.float32(m._update(128, False) * np.float32(256))) == 122
    # odds_table_[0] = log(0) = -inf: a voxel at probability 0 stays clamped at kMinProb = 0.1 -> uint8(25.6) = 25
    assert int(np.float32(m._update(0, True) * np.float32(256))) == 25
    # the clamps of Initialise (:50-51)
    c = pm.Mrvm(hit_prob=0.3, miss_prob=0.7)
    assert float(c.hit) == float(np.float32(0.501)) and float(c.miss) == float(np.float32(0.499))
    c = pm.Mrvm(hit_prob=0.99, miss_prob=0.01)
    assert float(c.hit) == float(np.float32(0.9)) and float(c.miss) == float(np.float32(0.1))
    # the C restatement's byte -> byte tables are the same function
    o = cref.Mrvm(hit_prob=0.55, miss_prob=0.48)
    hit, miss = o.tables()
    o.close()
    assert [int(np.float32(m._update(i, True) * np.float32(256))) & 0xff for i in range(256)] == list(hit)
    assert [int(np.float32(m._update(i, False) * np.float32(256))) & 0xff for i in range(256)] == list(miss)


def test_one_ray_then_a_shorter_one_by_hand():
    """Frame 1: one point at x = 0.45 -> voxel (4, 0, 0) is hit (140); the voxels before it do not exist, nothing is missed.
    Frame 2: a point at x = 0.85 -> its ray passes THROUGH (4, 0, 0), which exists and may be updated again: one miss
    from 140.  Frame 3: both points in one frame, the near one first: its end voxel is hit and its need_update flag stays
    false until the frame ends (:91, :123-125), so the far point's ray passes through it without a miss."""
    for M in (pm.Mrvm(), cref.Mrvm()):
        M.insert(np.array([[0.45, 0.05, 0.05, 7.9, 0]], np.float32), (0.05, 0.05, 0.05))
        keys, prob, mi, npts = M.dump()[:4]
        assert keys.tolist() == [[4, 0, 0]] and prob.tolist() == [140] and mi.tolist() == [7] and npts.tolist() == [1]
        M.insert(np.array([[0.85, 0.05, 0.05, 3.0, 0]], np.float32), (0.05, 0.05, 0.05))
        keys, prob, mi, npts = M.dump()[:4]
        assert keys.tolist() == [[4, 0, 0], [8, 0, 0]]
        ref = pm.Mrvm()
        assert prob.tolist() == [int(np.float32(ref._update(140, False) * np.float32(256))), 140]
        # a point landing in an end voxel of the same frame protects it from that frame's later rays only
        M.insert(np.array([[0.45, 0.05, 0.05, 1.0, 0], [0.85, 0.05, 0.05, 1.0, 0]], np.float32), (0.05, 0.05, 0.05))
        keys2, prob2 = M.dump()[:2]
        p4 = int(np.float32(ref._update(prob.tolist()[0], True) * np.float32(256)))        # hit by the first point, not missed by the second
        p8 = int(np.float32(ref._update(140, True) * np.float32(256)))
        assert prob2.tolist() == [p4, p8]
        if hasattr(M, "close"):
            M.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_and_python_restatements_agree(seed):
    rng = np.random.default_rng(seed)
    C = cref.Mrvm(high_resolution=0.2, hit_prob=0.6, miss_prob=0.45, z_offset=0.3, max_point_num_in_cell=3)
    P = pm.Mrvm(high_resolution=0.2, hit_prob=0.6, miss_prob=0.45, z_offset=0.3, max_point_num_in_cell=3)
    for frame in range(4):
        n = 150
        pts = np.zeros((n, 5), np.float32)
        d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts[:, :3] = d * rng.uniform(0.3, 4.0, (n, 1)) * [1, 1, 0.3]
        pts[:, 3] = rng.uniform(-3, 250, n)
        origin = rng.normal(0, 0.2, 3)
        C.insert(pts, origin); P.insert(pts, origin)
        kc, pc, mc, nc = C.dump()[:4]
        kp, pp, mp, npn = P.dump()
        assert np.array_equal(kc, kp) and np.array_equal(pc, pp) and np.array_equal(mc, mp) and np.array_equal(nc, npn), frame
    for thr in (0.5, 0.6, 0.7):
        for kw in (dict(), dict(average=True), dict(rgb=True), dict(average=True, rgb=True), dict(use_max_intensity=False), dict(average=True, use_max_intensity=False)):
            assert np.array_equal(_sorted_output(C.output(thr, **kw)), _sorted_output(P.output(thr, **kw))), (thr, kw)
    a = P.output(0.5, average=True); v = P.output(0.5)
    assert 0 < len(a) < len(v)                                                       # one averaged point per voxel
    assert set(np.unique(P.output(0.5, rgb=True)[:, 3])) <= set(float(g) for g in range(256))
    C.close()
