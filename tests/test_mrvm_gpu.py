"""static_map::MultiResolutionVoxelMap on the device (csrc/smhip_mrvm.hip) against the oracle's restatement of the
reference's insert loop in point order (oracle/csrc/smref_mrvm.c, /root/reference/builder/multi_resolution_voxel_map.cc:59-131):
the whole map -- voxel set, probability bytes, max intensities, stored points -- must be IDENTICAL after every cloud."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _world_scans(n_scans, n_points, seed=5):
    from staticmapping_amd import synth
    poses = synth.drive_poses(n_scans, seed=seed, speed=8.0)
    scene = synth.make_drive_scene(poses, seed=seed)
    out = []
    for k, P in enumerate(poses):
        s = synth.velodyne_scan(synth.scene_near(scene, P[:3, 3]), P, seed=10 + k, n_points=n_points)
        w = (s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]).astype(np.float32)       # ApplyTransformToOutput(GlobalPose), map_builder.cc:842-843
        pts = np.concatenate([w, np.round(s[:, 3:4] * 255), (np.arange(len(s), dtype=np.float32) / len(s))[:, None]], axis=1).astype(np.float32)
        out.append((pts, P[:3, 3].astype(np.float32)))
    return out


def _assert_same_map(dev, ora):
    kd, pd, md, nd, qd = dev.dump()
    ko, po, mo, no, qo = ora.dump()
    assert len(kd) == len(ko)
    assert np.array_equal(kd, ko)
    assert np.array_equal(pd, po), int((pd != po).sum())
    assert np.array_equal(md, mo)
    assert np.array_equal(nd, no)
    assert np.array_equal(qd, qo)


@pytest.mark.parametrize("settings", [dict(), dict(high_resolution=0.25, hit_prob=0.7, miss_prob=0.4, max_point_num_in_cell=3, z_offset=0.3)])
def test_map_equals_the_reference_loop_in_point_order(settings):
    import staticmapping_amd as sm
    from oracle import cref
    scans = _world_scans(4, 30_000)
    dev = sm.MultiResolutionVoxelMapHip(table_log2=20, max_cloud_points=30_000, **settings)
    ora = cref.Mrvm(**{k: v for k, v in settings.items()})
    for pts, origin in scans:
        dev.insert_point_cloud(pts, origin)
        ora.insert(pts, origin)
        _assert_same_map(dev, ora)                               # after EVERY cloud: misses of later clouds hit earlier voxels
    for thr in (0.6, 0.52, 0.9):
        # both OutputToPointCloud overloads, with and without MrvmSettings::output_average (.cc:125-216): the same rows, bit for bit
        for kw in (dict(), dict(average=True), dict(rgb=True), dict(average=True, rgb=True)):
            a = dev.output_to_point_cloud(thr, **kw)
            b = ora.output(thr, **kw)
            assert len(a) == len(b), (thr, kw)
            sa = a[np.lexsort(a.T[::-1])]; sb = b[np.lexsort(b.T[::-1])]
            assert np.array_equal(sa, sb), (thr, kw)
    assert len(dev.output_to_point_cloud(0.52, average=True)) < len(dev.output_to_point_cloud(0.52))     # one row per voxel
    assert dev.voxel_count() == len(ora.dump()[0])
    dev.close(); ora.close()


def test_repeated_scan_saturates_and_clears():
    """The same cloud over and over drives its end voxels to the clamp (0.9 -> byte 230); a second sensor position whose rays pass
    through them drives them down again: both directions of the update, through many applications of the byte maps."""
    import staticmapping_amd as sm
    from oracle import cref
    (pts, origin), (pts2, origin2) = _world_scans(2, 8_000, seed=9)
    dev = sm.MultiResolutionVoxelMapHip(table_log2=18, max_cloud_points=8_000)
    ora = cref.Mrvm()
    for _ in range(25):
        dev.insert_point_cloud(pts, origin); ora.insert(pts, origin)
    _assert_same_map(dev, ora)
    assert dev.dump()[1].max() == 230
    far = pts2.copy(); far[:, :3] = origin2 + (pts[:, :3] - origin) * 1.5          # rays through the first cloud's end voxels
    for _ in range(30):
        dev.insert_point_cloud(far, origin); ora.insert(far, origin)
    _assert_same_map(dev, ora)
    dev.close(); ora.close()


def test_edge_cases():
    import staticmapping_amd as sm
    from oracle import cref
    dev = sm.MultiResolutionVoxelMapHip(table_log2=12, max_cloud_points=4096, max_table_log2=12)   # (a table that may not grow)
    ora = cref.Mrvm()
    with pytest.raises(sm.SmhipError):
        dev.insert_point_cloud(np.zeros((0, 5), np.float32), [0, 0, 0])                # "cloud is empty."
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(0, 3, (500, 3)), rng.uniform(0, 255, (500, 1)), np.zeros((500, 1))], axis=1).astype(np.float32)
    pts[7, 0] = np.nan; pts[9, 2] = np.inf                                            # skipped, like every other non-finite point
    pts[11, :3] = [0.05, 0.05, 0.05]                                                  # a point in the origin's own voxel: a ray of one voxel
    pts[12] = pts[13]                                                                 # duplicates share a voxel
    for origin in ([0, 0, 0], [0.5, -0.25, 0.1]):
        dev.insert_point_cloud(pts, origin); ora.insert(pts, origin)
        _assert_same_map(dev, ora)
    # a point beyond +-2^20 voxels is skipped, the rest of the cloud is applied, and the NEXT insert knows nothing of it
    far = pts.copy(); far[20, 0] = 2.0e5                                               # 2e6 voxels of 0.1 m
    dev.insert_point_cloud(far, [0, 0, 0])
    assert dev.last_skipped == 1 and "skipped" in dev.last_warning
    keep = np.ones(len(far), bool); keep[20] = False
    ora.insert(far[keep], [0, 0, 0])                                                   # (the skipped point's ray is not cast either)
    _assert_same_map(dev, ora)
    dev.insert_point_cloud(pts, [0.1, 0, 0]); ora.insert(pts, [0.1, 0, 0])
    assert dev.last_skipped == 0 and dev.last_warning == ""
    _assert_same_map(dev, ora)
    # a non-finite or far-away origin is refused before anything is touched
    for bad in ([np.nan, 0, 0], [0, 3.0e5, 0]):
        with pytest.raises(sm.SmhipError):
            dev.insert_point_cloud(pts, bad)
    _assert_same_map(dev, ora)
    # 4096-slot table: past 70 % the insert is applied and says so; once a voxel does not fit the map reports it from then on
    more = np.concatenate([rng.normal(0, 30, (2600, 3)), np.zeros((2600, 2))], axis=1).astype(np.float32)
    dev.insert_point_cloud(more, [0, 0, 0]); ora.insert(more, [0, 0, 0])
    assert "70 %" in dev.last_warning
    _assert_same_map(dev, ora)
    with pytest.raises(sm.SmhipError):
        big = np.concatenate([rng.normal(0, 30, (4096, 3)), np.zeros((4096, 2))], axis=1).astype(np.float32)
        dev.insert_point_cloud(big, [0, 0, 0])
    with pytest.raises(sm.SmhipError):                                                 # sticky: a voxel was lost
        dev.insert_point_cloud(pts, [0, 0, 0])
    dev.close(); ora.close()


def test_table_grows_like_the_reference_map():
    """The reference's map is a std::map and grows without bound (multi_resolution_voxel_map.h).  A device table that starts far
    too small (1 024 slots) doubles between inserts; after every cloud the whole map still equals the reference loop's."""
    import staticmapping_amd as sm
    from oracle import cref
    scans = _world_scans(5, 20_000)
    dev = sm.MultiResolutionVoxelMapHip(table_log2=10, max_cloud_points=20_000)
    ora = cref.Mrvm()
    sizes = []
    for pts, origin in scans:
        dev.insert_point_cloud(pts, origin); ora.insert(pts, origin)
        assert dev.last_warning == ""
        _assert_same_map(dev, ora)
        sizes.append(dev.table_log2)
    assert sizes[0] > 10 and sizes[-1] >= sizes[0] and dev.voxel_count() * 2 <= (1 << sizes[-1]) + 20_000, sizes
    a = dev.output_to_point_cloud(0.6); b = ora.output(0.6)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    # a limit below what the map needs: the old behaviour (warning, then a lost voxel reported from then on)
    small = sm.MultiResolutionVoxelMapHip(table_log2=10, max_cloud_points=20_000, max_table_log2=11)
    with pytest.raises(sm.SmhipError):
        small.insert_point_cloud(scans[0][0], scans[0][1])
    assert small.table_log2 == 11
    small.close(); dev.close(); ora.close()


def test_cpp_mirror_drives_the_map_like_the_map_builder(tmp_path):
    """include/smhip/mrvm.h through a small C++ program (tests/cpp/test_mrvm.cc) against the oracle."""
    import json, os, subprocess
    from staticmapping_amd import build
    from oracle import cref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = build.build()
    exe = os.path.join(root, "tests", "cpp", "_build", "test_mrvm")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "test_mrvm.cc"), "-o", exe,
                           "-L", os.path.dirname(lib), "-lsmhip", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    scans = _world_scans(3, 20_000, seed=3)
    ora = cref.Mrvm()
    args = []
    for k, (pts, origin) in enumerate(scans):
        f = tmp_path / f"c{k}.bin"
        pts.tofile(f)
        args += [str(f)] + [repr(float(v)) for v in origin]
        ora.insert(pts, origin)
    res = json.loads(subprocess.check_output([exe] + args, text=True, timeout=120).strip().splitlines()[-1])
    out = ora.output(0.6)
    assert res["empty_refused"] and res["voxels"] == len(ora.dump()[0]) and res["output_points"] == len(out)
    want = float((out[:, 0].astype(np.float64) + 2.0 * out[:, 1] + 3.0 * out[:, 2] + 0.001 * out[:, 3]).sum())
    assert abs(res["checksum"] - want) <= 1e-6 * max(1.0, abs(want))
    cs = lambda o: float((o[:, 0].astype(np.float64) + 2.0 * o[:, 1] + 3.0 * o[:, 2] + 0.001 * o[:, 3]).sum())
    rgb = ora.output(0.6, rgb=True); avg = ora.output(0.6, average=True)
    assert res["grey"] and res["rgb_points"] == len(rgb) and abs(res["rgb_checksum"] - cs(rgb)) <= 1e-6 * max(1.0, abs(cs(rgb)))
    assert res["average_points"] == len(avg) and abs(res["average_checksum"] - cs(avg)) <= 1e-6 * max(1.0, abs(cs(avg)))
    ora.close()
