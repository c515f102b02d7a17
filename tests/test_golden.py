"""Committed fixtures (tests/golden/*.json, made by tests/golden/make_golden.py from the numpy oracle):
the C restatement, the numpy restatement and -- on the GPU box -- the HIP path must reproduce them."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _icp_inputs():
    from staticmapping_amd import synth
    from oracle import icp_fast as o
    tgt, src, T = synth.three_planes_pair(2000, seed=11, sigma=0.01)
    q, n, _ = o.calculate_normals(tgt[:, :3].astype(np.float64))
    return src, q, n


def _ndt_inputs():
    from staticmapping_amd import synth
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.8 * k, 0.0, 0.0)) for k in range(3)]
    scans = [synth.velodyne_scan(scene, P, seed=20 + k, n_points=8000) for k, P in enumerate(poses)]
    tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:2], poses[:2])]).astype(np.float32)
    G = poses[2].copy(); G[0, 3] -= 0.25
    return scans[2], tgt, G


def test_icp_golden_numpy_and_c():
    from oracle import icp_fast as o, cref
    g = json.load(open(os.path.join(HERE, "golden", "icp_three_planes_2000.json")))
    src, q, n = _icp_inputs()
    assert len(q) == g["target_points"]
    trace = []
    R, score, it = o.icp_fast_align(src[:, :3].astype(np.float64), q, n, trace=trace)
    assert it == g["iterations"] and abs(score - g["score"]) < 1e-12
    assert np.allclose(R, np.array(g["result"]), atol=1e-10)
    assert trace[0]["limit"] == pytest.approx(g["first_limit_d2"], rel=1e-12)
    assert int(trace[0]["keep"].sum()) == g["first_kept"]
    r = cref.icp_fast_align(src[:, :3].astype(np.float64), q, n)
    assert r["iterations"] == g["iterations"]
    assert np.allclose(r["result"], np.array(g["result"]), atol=1e-9)


def test_ndt_golden_numpy():
    from oracle import ndt as ondt
    g = json.load(open(os.path.join(HERE, "golden", "ndt_two_scans_8000.json")))
    src, tgt, G = _ndt_inputs()
    r = ondt.ndt_align(src, tgt, guess=G)
    assert r["iterations"] == g["iterations"] and r["derivative_calls"] == g["derivative_calls"]
    assert np.allclose(r["result"], np.array(g["result"]), atol=1e-6)
    assert abs(r["score"] - g["score"]) <= 1e-6 * g["score"]


@pytest.mark.gpu
def test_icp_golden_gpu():
    import staticmapping_amd as sm
    g = json.load(open(os.path.join(HERE, "golden", "icp_three_planes_2000.json")))
    src, q, n = _icp_inputs()
    m = sm.IcpFastHip(max_source_points=4096, max_target_points=4096)
    m.set_input_source(src); m.set_input_target(q, n)
    ok, R = m.align()
    da, dt = sm.se3_error(R, np.array(g["result"]))
    assert da < 1e-4 and dt < 1e-3
    assert m.last_stats[0]["iterations"] == g["iterations"]
    assert abs(m.get_fitness_score() - g["score"]) < 1e-4
    m.close()


@pytest.mark.gpu
def test_ndt_golden_gpu():
    import staticmapping_amd as sm
    g = json.load(open(os.path.join(HERE, "golden", "ndt_two_scans_8000.json")))
    src, tgt, G = _ndt_inputs()
    tgt4 = np.concatenate([tgt, np.zeros((len(tgt), 1), np.float32)], axis=1)
    m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt4))
    m.set_input_source(src); m.set_input_target(tgt4)
    ok, R = m.align(G)
    da, dt = sm.se3_error(R, np.array(g["result"]))
    assert da < 1e-4 and dt < 1e-3
    assert m.last_ndt_stats["iterations"] == g["iterations"]
    assert abs(m.get_fitness_score() - g["score"]) <= 1e-3 * g["score"]
    m.close()


@pytest.mark.gpu
def test_reference_transform_known_answer_through_the_device():
    """builder/data/test/test_cloud_types.cc:166-187 (TransformPointAndCloud): the point (10, 0, 30) under the identity
    stays put and under Vector6ToTransform(0,0,0,0,0,pi) goes to (-10, -0, 30).  The device applies the caller's
    transform inside FindClosests (ApplyTransform is fused), so the known answer is replayed through
    it: the transformed point must land on a target point placed at the reference's expected coordinates."""
    import staticmapping_amd as sm
    src = np.array([[10.0, 0.0, 30.0], [1.0, 2.0, 3.0], [-4.0, 5.0, 6.0], [7.0, -8.0, 9.0]])
    tgt = np.array([[-10.0, 0.0, 30.0], [10.0, 0.0, 30.0], [50.0, 50.0, 50.0], [-1.0, -2.0, 3.0], [4.0, -5.0, 6.0], [-7.0, 8.0, 9.0]])
    nrm = np.tile([0.0, 0.0, 1.0], (len(tgt), 1))
    m = sm.IcpFastHip(pair_slots=1, max_source_points=16, max_target_points=16)
    m.set_input_source(src); m.set_input_target(tgt, nrm)
    ids, d2 = m.find_closests(np.eye(4), len(src))
    assert ids[0] == 1 and d2[0] == 0.0                                     # identity: (10, 0, 30)
    Rz = np.eye(4); Rz[0, 0] = Rz[1, 1] = np.cos(np.pi); Rz[0, 1] = -np.sin(np.pi); Rz[1, 0] = np.sin(np.pi)
    ids, d2 = m.find_closests(Rz, len(src))
    assert list(ids) == [0, 3, 4, 5]                                        # (-x, -y, z) of every source point
    assert np.all(d2 < 1e-10)                                               # BOOST_CHECK_DOUBLE_EQUAL tolerance squared
    m.close()


def _load_make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_filters_and_ndt_gicp_golden_oracle():
    import hashlib
    from oracle import filters as of, ndt_gicp as ong
    mg = _load_make_golden()
    g = json.load(open(os.path.join(HERE, "golden", "filters_kitti_chain_12000.json")))
    rows, chain = mg.filter_inputs()
    out3, src3 = of.run_chain(of.with_factor(rows), chain[:3])
    out4, _ = of.run_chain(of.with_factor(rows), chain)
    assert len(out3) == g["n_after_sampler"] and hashlib.sha256(out3.tobytes()).hexdigest() == g["sampler_sha256"]
    assert hashlib.sha256(src3.tobytes()).hexdigest() == g["index_sha256"]
    assert len(out4) == g["n_after_voxel_grid"] and hashlib.sha256(out4.tobytes()).hexdigest() == g["voxel_sha256"]
    g = json.load(open(os.path.join(HERE, "golden", "ndt_gicp_cfg2_12000.json")))
    src, tgt, G, T = mg.ndt_gicp_inputs()
    ds = ong.approximate_voxel_grid(src, 0.2)
    assert hashlib.sha256(ds.tobytes()).hexdigest() == g["downsampled_source_sha256"]
    r = ong.ndt_gicp_align(src, tgt, G)
    assert (r["n_source"], r["n_target"], r["ok"]) == (g["n_source"], g["n_target"], g["ok"])
    assert r["ndt"]["iterations"] == g["ndt_iterations"] and r["gicp"]["iterations"] == g["gicp_iterations"]
    assert np.allclose(r["result"], np.array(g["result"]), atol=1e-6)


@pytest.mark.gpu
def test_filters_and_ndt_gicp_golden_gpu():
    import hashlib
    import staticmapping_amd as sm
    from staticmapping_amd import filters as df
    mg = _load_make_golden()
    g = json.load(open(os.path.join(HERE, "golden", "filters_kitti_chain_12000.json")))
    rows, ochain = mg.filter_inputs()
    chain = [df.make_filter("Range", min_range=5.0), df.make_filter("AxisRange", min=-2.0),
             df.make_filter("RandomSampler", sampling_rate=0.5, seed=21), df.make_filter("VoxelGrid", voxel_size=0.3)]
    m = sm.NdtGicpHip(max_source_points=16384, max_target_points=16384)
    out3, src3 = df.run_chain(m, rows, chain[:3])
    out4, _ = df.run_chain(m, rows, chain)
    assert hashlib.sha256(out3.tobytes()).hexdigest() == g["sampler_sha256"]
    assert hashlib.sha256(src3.tobytes()).hexdigest() == g["index_sha256"]
    assert hashlib.sha256(out4.tobytes()).hexdigest() == g["voxel_sha256"]
    g = json.load(open(os.path.join(HERE, "golden", "ndt_gicp_cfg2_12000.json")))
    src, tgt, G, T = mg.ndt_gicp_inputs()
    m.set_input_source(src); m.set_input_target(tgt)
    ok, R = m.align(G)
    st = m.last_gicp_stats
    assert ok == g["ok"] and (st["n_source"], st["n_target"]) == (g["n_source"], g["n_target"])
    assert hashlib.sha256(m.get_downsampled(0).tobytes()).hexdigest() == g["downsampled_source_sha256"]
    # the GICP outer loop stops on a threshold (max |delta| < 1 in units of 1e-3 rad / 0.5 mm): last-bit differences in
    # the covariances can move the stop by one iteration; the pose tolerance below is what counts
    assert st["ndt_iterations"] == g["ndt_iterations"] and abs(st["gicp_iterations"] - g["gicp_iterations"]) <= 2
    da, dt = sm.se3_error(R, np.array(g["result"]))
    assert da < 3e-3 and dt < 5e-2, (da, dt)            # GICP's own repeatability, see tests/test_ndt_gicp_gpu.py
    assert abs(m.get_fitness_score() - g["score"]) < 2e-2
    assert abs(st["ndt_score"] - g["ndt_score"]) < 1e-3 * g["ndt_score"]
    m.close()
