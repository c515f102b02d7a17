"""BASELINE config #4 end to end on one GPU: a synthetic drive, scan-to-scan ICP, everything included
(host->device uploads, Morton ordering, device CalculateNormals, alignment with early exit, read-back)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth, kitti, shard
K = int(sys.argv[1]) if len(sys.argv) > 1 else 17
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.03 * k, 0.0), rpy_deg=(0, 0, 1.0 * k)) for k in range(K)]
t = time.time()
scans = [synth.velodyne_scan(scene, P, seed=500 + k, n_points=120000) for k, P in enumerate(poses)]
print(f"generated {K} scans in {time.time()-t:.1f} s", flush=True)
rel_true = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(K - 1)]
guesses = []
for T in rel_true:
    G = np.eye(4); G[:3, 3] = 0.75 * T[:3, 3]; guesses.append(G)
for batch in (16, 64):
    m = sm.IcpFastHip(pair_slots=batch + 1, max_source_points=120000, max_target_points=120000 // 4 + 64, max_iteration=100, early_exit=1)
    kitti.scan_to_scan_sequence(scans[:3], m, batch=batch, guesses=guesses)      # warm-up (workspace allocation)
    t = time.time()
    idx, T, sc, it = kitti.scan_to_scan_sequence(scans, m, batch=batch, guesses=guesses)
    dt = time.time() - t
    errs = [sm.se3_error(T[i], rel_true[i]) for i in range(len(idx))]
    print(f"batch={batch}: {len(idx)} pairs in {dt*1e3:.1f} ms = {len(idx)/dt:.1f} pairs/s end to end; iterations {it.min()}..{it.max()}; "
          f"worst err rot {max(e[0] for e in errs):.2e} rad trans {max(e[1] for e in errs):.2e} m; "
          f"trajectory end error {np.linalg.norm(shard.chain_poses(T)[-1][:3,3] - (np.linalg.inv(poses[0]) @ poses[-1])[:3,3]):.3f} m")
    m.close()
