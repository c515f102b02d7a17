"""IcpFast at submap scale (SubmapPairMatch / CloseLoop sized clouds): two submaps of `n_scans`
merged 120k-pt scans each; target through the device CalculateNormals."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.3 * k)) for k in range(2 * n_scans)]
scans = [synth.velodyne_scan(scene, P, seed=100 + k, n_points=120000) for k, P in enumerate(poses)]
def merge(ids):
    base = np.linalg.inv(poses[ids[0]])
    return np.concatenate([scans[k][:, :3].astype(np.float64) @ (base @ poses[k])[:3, :3].T + (base @ poses[k])[:3, 3] for k in ids]).astype(np.float32)
tgt, src = merge(range(n_scans)), merge(range(n_scans // 2, n_scans + n_scans // 2))
T = np.linalg.inv(poses[0]) @ poses[n_scans // 2]
G = T.copy(); G[0, 3] -= 0.3
m = sm.IcpFastHip(pair_slots=2, max_source_points=len(src), max_target_points=len(tgt), max_iteration=100, early_exit=1)
t = time.time(); m.set_input_source(np.ascontiguousarray(np.c_[src, np.zeros(len(src), np.float32)])); t_src = time.time() - t
t = time.time(); nt = m.prepare_target(np.ascontiguousarray(np.c_[tgt, np.zeros(len(tgt), np.float32)])); t_tgt = time.time() - t
ok, R = m.align(G)
t = time.time()
for _ in range(3): ok, R = m.align(G)
dt = (time.time() - t) / 3
print(f"submaps {len(src)} vs {len(tgt)} (prepared target {nt}): set_source {t_src*1e3:.1f} ms, prepare_target {t_tgt*1e3:.1f} ms, "
      f"align {dt*1e3:.2f} ms, iterations {m.last_stats[0]['iterations']} score {m.get_fitness_score():.4f} err {sm.se3_error(R, T)} stats {m.last_stats[0]}")
