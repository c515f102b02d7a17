"""NDT timing probe at BASELINE config #3 scale: 120k-pt scan vs 500k-pt submap, 1.0 m voxels."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.5 * k)) for k in range(6)]
scans = [synth.velodyne_scan(scene, P, seed=40 + k, n_points=120000) for k, P in enumerate(poses)]
tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:5], poses[:5])])
rng = np.random.default_rng(4)
tgt = tgt[rng.choice(len(tgt), size=n_t, replace=False)]
tgt = np.concatenate([tgt, np.zeros((len(tgt), 1))], axis=1).astype(np.float32)
src = scans[5]; T = poses[5]
G = T.copy(); G[0, 3] -= 0.3; 
m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
m.set_input_source(src); m.set_input_target(tgt)
for cache in (False, True):           # False = the reference's behaviour: voxel table + fitness search structure rebuilt in every Align
    m.set_target_cache(cache)
    m.align(G)
    t = time.time(); reps = 5
    for _ in range(reps): ok, R = m.align(G)
    dt = (time.time() - t) / reps
    print(f"NDT {len(src)} vs {len(tgt)} target_cache={int(cache)}: {dt*1e3:.2f} ms/align  {1/dt:.1f} align/s stats={m.last_ndt_stats} score={m.get_fitness_score():.5f} err={sm.se3_error(R, T)}")
