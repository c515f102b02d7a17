"""Write a synthetic drive (SURVEY.md §8(d) cfg 4: speed 8 m/s, 10 Hz, yaw rate U[-0.2, 0.2] rad/s, seed 5) as a KITTI
odometry sequence: <dir>/%010d.bin (float32 rows x y z reflectance, ros_node/kitti_reader.cc:91-121) and
<dir>/../<name>_truth.txt (kitti_pose.txt format, builder/map_builder.cc:626-641) with the generating poses.
Usage: python tools/make_drive.py OUT_DIR N_SCANS [N_POINTS=120000] [--cuda]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from staticmapping_amd import kitti, shard, synth  # noqa: E402


def main():
    out, n = sys.argv[1], int(sys.argv[2])
    n_points = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 120_000
    device = None
    if "--cuda" in sys.argv:
        import torch
        device = torch.device("cuda:0")
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    poses = synth.drive_poses(n, seed=5, speed=8.0, hz=10.0, yaw_rate_max=0.2)
    scene = synth.make_drive_scene(poses, seed=5)
    for k, P in enumerate(poses):
        near = synth.scene_near(scene, P[:3, 3])
        kitti.write_bin(kitti.scan_path(out, k), synth.velodyne_scan(near, P, seed=1000 + k, n_points=n_points, device=device))
    base = np.linalg.inv(poses[0])
    kitti.write_poses(os.path.join(os.path.dirname(os.path.abspath(out)), os.path.basename(os.path.abspath(out)) + "_truth.txt"),
                      np.stack([base @ P for P in poses]))
    print(f"make_drive: {n} scans x {n_points} points -> {out} in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
