"""BASELINE config #4 in miniature: a synthetic drive, scan-to-scan pairs in one batch, chained trajectory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_synthetic_drive_scan_to_scan(tmp_path):
    import staticmapping_amd as sm
    from staticmapping_amd import synth, kitti, shard
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.5 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 1.0 * k)) for k in range(7)]
    for k, P in enumerate(poses):
        kitti.write_bin(kitti.scan_path(str(tmp_path), k), synth.velodyne_scan(scene, P, seed=70 + k, n_points=20000))
    files = kitti.list_scans(str(tmp_path))
    scans = [(lambda f=f: kitti.read_bin(f)) for f in files]              # lazy, reference reader semantics
    rel_true = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(6)]
    guesses = []
    for T in rel_true:                                                    # constant-velocity-like prediction
        G = np.eye(4); G[:3, 3] = 0.75 * T[:3, 3]; guesses.append(G)
    m = sm.IcpFastHip(pair_slots=3, max_source_points=20000, max_target_points=8192)
    got = {}
    for rank in range(2):                                                 # two "ranks" on one GPU: the shard logic
        idx, T, sc, it = kitti.scan_to_scan_sequence(scans, m, batch=3, guesses=guesses, rank=rank, world=2)
        for i, Ti in zip(idx, T):
            got[i] = Ti
    m.close()
    assert sorted(got) == list(range(6))
    rel = np.stack([got[i] for i in range(6)])
    for i in range(6):
        da, dt = sm.se3_error(rel[i], rel_true[i])
        assert da < 2e-3 and dt < 2e-2, (i, da, dt)
    traj = shard.chain_poses(rel)
    assert np.linalg.norm(traj[-1][:3, 3] - (np.linalg.inv(poses[0]) @ poses[-1])[:3, 3]) < 0.05
    kitti.write_poses(str(tmp_path / "kitti_pose.txt"), traj)
    assert len(open(tmp_path / "kitti_pose.txt").read().splitlines()) == 7


def test_synthetic_drive_with_the_kitti_filter_chain(tmp_path):
    """The same drive with config/lidar_only_kitti.xml's <filters> chain applied on the device to every scan:
    filtered sources and filtered targets never return to the host between load and Align."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth, kitti, filters as df
    from oracle import filters as of, cref
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.5 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 1.0 * k)) for k in range(5)]
    scans = [synth.velodyne_scan(scene, P, seed=80 + k, n_points=30000) for k, P in enumerate(poses)]
    rel_true = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(4)]
    guesses = []
    for T in rel_true:
        G = np.eye(4); G[:3, 3] = 0.75 * T[:3, 3]; guesses.append(G)
    chain = [df.make_filter("Range", min_range=5.0), df.make_filter("AxisRange", min=-2.0),
             df.make_filter("RandomSampler", sampling_rate=0.5, seed=11)]
    m = sm.IcpFastHip(pair_slots=5, max_source_points=30000, max_target_points=30000, max_iteration=30, early_exit=1)
    idx, T, sc, it = kitti.scan_to_scan_sequence(scans, m, batch=4, guesses=guesses, filter_chain=chain)
    assert list(idx) == [0, 1, 2, 3]
    # the oracle on the same filtered clouds: filters (pinned oracle) -> CalculateNormals -> IcpFast restatement
    ochain = [dict(of.default(of.RANGE), min_range=5.0), dict(of.default(of.AXIS_RANGE), min=-2.0),
              dict(of.default(of.RANDOM_SAMPLER), sampling_rate=0.5, seed=11)]
    for i in range(4):
        fs, _ = of.run_chain(of.with_factor(scans[i + 1][:, :4]), ochain)
        ft, _ = of.run_chain(of.with_factor(scans[i][:, :4]), ochain)
        q, n = sm.calculate_normals(ft[:, :3].astype(np.float64))
        ref = cref.icp_fast_align(fs[:, :3].astype(np.float64), q, n, guess=guesses[i], max_iteration=30)
        da, dt = sm.se3_error(T[i], ref["result"])
        assert da < 1e-4 and dt < 1e-3, (i, da, dt)
        da, dt = sm.se3_error(T[i], rel_true[i])
        assert da < 3e-3 and dt < 3e-2, (i, da, dt)
    m.close()


@pytest.mark.gpu
def test_batched_upload_equals_single_uploads(velo20k):
    """smhip_set_sources_f32_batch (one Morton ordering for many scans; here from pageable memory, the driver test covers the
    page-locked path) leaves every slot with exactly the source a single smhip_set_source_f32 leaves: same alignment bits."""
    import numpy as np
    import staticmapping_amd as sm
    rng = np.random.default_rng(4)
    base = np.ascontiguousarray(velo20k["src"][:, :4], dtype=np.float32)
    clouds = [np.ascontiguousarray(base[rng.permutation(len(base))[:n]]) for n in (20000, 12345, 1500, 19999, 16000)]
    guesses = [velo20k["guess"]] * len(clouds)
    out = []
    for batched in (False, True):
        m = sm.IcpFastHip(pair_slots=len(clouds), max_source_points=20000, max_target_points=len(velo20k["q"]), max_iteration=12, early_exit=0)
        if batched:
            m.set_input_sources_batch(clouds, list(range(len(clouds))))
        else:
            for s, c in enumerate(clouds):
                m.set_input_source(c, slot=s)
        for s in range(len(clouds)):
            m.set_input_target(velo20k["q"], velo20k["n"], slot=s)
        R, sc, st = m.align_batch(len(clouds), guesses)
        m.close()
        out.append((np.asarray(R).tobytes(), [x["kept"] for x in st]))
    assert out[0] == out[1]
