"""NdtWithGicp timing probe at BASELINE config #5 scale: 120k-pt scan vs a 2M-pt accumulated submap (20 merged
scans), ApproximateVoxelGrid 0.2 m -> pcl NDT -> pcl GICP.  Usage: gicp_probe.py [n_target] [n_scans]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.3 * k)) for k in range(n_scans + 1)]
scans = [synth.velodyne_scan(scene, P, seed=60 + k, n_points=120000) for k, P in enumerate(poses)]
tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:n_scans], poses[:n_scans])])
rng = np.random.default_rng(6)
if n_t < len(tgt):
    tgt = tgt[np.sort(rng.choice(len(tgt), size=n_t, replace=False))]
tgt = np.ascontiguousarray(tgt.astype(np.float32))
src = np.ascontiguousarray(scans[n_scans][:, :3]); T = poses[n_scans]
G = T.copy(); G[0, 3] -= 0.3
m = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
t = time.time(); m.set_input_source(src); m.set_input_target(tgt); t_up = time.time() - t
reps = 3
for cache in (False, True):
    m.set_target_cache(cache)
    ok, R = m.align(G)
    t = time.time()
    for _ in range(reps): ok, R = m.align(G)
    dt = (time.time() - t) / reps
    print(f"NdtWithGicp {len(src)} vs {len(tgt)} target_cache={int(cache)}: upload {t_up*1e3:.1f} ms, {dt*1e3:.2f} ms/align ok={ok} stats={m.last_gicp_stats} "
          f"score={m.get_fitness_score():.5f} err={sm.se3_error(R, T)}")
m.set_target_cache(False)
for name, kw in (("filter+gicp (use_ndt=0)", dict(use_ndt=0)), ("filter+ndt+1 gicp iteration", dict(use_ndt=1, gicp_max_iterations=1))):
    m.set_gicp_options(**kw)
    m.align(G)
    t = time.time()
    for _ in range(reps): m.align(G)
    print(f"  {name}: {(time.time() - t) / reps * 1e3:.2f} ms  stats={m.last_gicp_stats}")
