import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
n_scans = 20
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.3 * k)) for k in range(n_scans + 1)]
scans = [synth.velodyne_scan(scene, P, seed=60 + k, n_points=120000) for k, P in enumerate(poses)]
tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:n_scans], poses[:n_scans])])
rng = np.random.default_rng(6)
tgt = np.ascontiguousarray(tgt[np.sort(rng.choice(len(tgt), size=2_000_000, replace=False))].astype(np.float32))
src = np.ascontiguousarray(scans[n_scans][:, :3]); T = poses[n_scans]
G = T.copy(); G[0, 3] -= 0.3
m = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
m.set_input_source(src); m.set_input_target(tgt)
ref = None
for cell in (0.0, 0.4, 0.6, 0.8, 1.0, 1.5):
    m.set_gicp_options(gicp_search_cell=cell)
    ok, R = m.align(G)
    t = time.time()
    for _ in range(2): ok, R = m.align(G)
    dt = (time.time() - t) / 2
    if ref is None: ref = R
    print(f"cell {cell}: {dt*1e3:.1f} ms same={np.array_equal(R, ref)}")
