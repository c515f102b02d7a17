import json, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth, kitti
scene = synth.make_scene(0)
n = 9
poses = [synth.make_pose(t=(0.3 * k, 0.01 * k, 0.0), rpy_deg=(0, 0, 0.4 * k)) for k in range(n)]
d = tempfile.mkdtemp()
for k, P in enumerate(poses):
    kitti.write_bin(kitti.scan_path(d, k), synth.velodyne_scan(scene, P, seed=120 + k, n_points=30000))
out = subprocess.check_output([os.path.join(os.path.dirname(__file__), "..", "tests", "cpp", "_build", "test_front_end"), str(n), d, "3.0", "0.1"], text=True)
frames = json.loads(out.strip().splitlines()[-1])["frames"]
for k, f in enumerate(frames):
    P = np.array(f["pose"]).reshape(4, 4)
    print(k, f["key"], round(f["score"], 4), np.round(P[:3, 3], 3), "true", np.round((np.linalg.inv(poses[0]) @ poses[k])[:3, 3], 3))
