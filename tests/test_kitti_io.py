import numpy as np

from staticmapping_amd import kitti, shard


def test_bin_round_trip_and_intensity_scale(tmp_path):
    pts = np.random.default_rng(0).uniform(-50, 50, (1234, 4)).astype(np.float32)
    pts[:, 3] = np.linspace(0, 1, 1234, dtype=np.float32)
    p = tmp_path / "0000000007.bin"
    kitti.write_bin(str(p), pts)
    assert kitti.scan_path(str(tmp_path), 7) == str(p)
    back = kitti.read_bin(str(p))
    assert back.shape == (1234, 4) and np.array_equal(back[:, :3], pts[:, :3])
    assert np.allclose(back[:, 3], pts[:, 3] * 255.0)                     # kitti_reader.cc:110
    assert kitti.list_scans(str(tmp_path)) == [str(p)]


def test_bin_reader_caps_like_the_reference(tmp_path):
    big = np.zeros((260_000, 4), dtype=np.float32)                        # 1.04 M floats > the 1 M buffer
    p = tmp_path / "big.bin"
    kitti.write_bin(str(p), big)
    assert kitti.read_bin(str(p)).shape == (250_000, 4)                   # kitti_reader.cc:93,104


def test_pose_file_format(tmp_path):
    T = np.eye(4); T[:3, 3] = (0.8, 0.05, 0.01)
    poses = shard.chain_poses(np.stack([T] * 5))
    f = tmp_path / "kitti_pose.txt"
    kitti.write_poses(str(f), poses)
    lines = open(f).read().strip().splitlines()
    assert len(lines) == 6 and all(len(l.split()) == 12 for l in lines)   # map_builder.cc:626-641
    assert np.allclose(kitti.read_poses(str(f)), poses, atol=1e-7)
