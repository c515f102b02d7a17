import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
from oracle import ndt_gicp as ong
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene = synth.make_scene(0)
poses = [synth.make_pose(t=(0.8 * k, 0.02 * k, 0.0), rpy_deg=(0, 0, 0.3 * k)) for k in range(n_scans + 1)]
scans = [synth.velodyne_scan(scene, P, seed=60 + k, n_points=120000) for k, P in enumerate(poses)]
tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:n_scans], poses[:n_scans])])
tgt = np.ascontiguousarray(tgt.astype(np.float32))
src = np.ascontiguousarray(scans[n_scans][:, :3]); T = poses[n_scans]
G = T.copy(); G[0, 3] -= 0.3
m = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
m.set_input_source(src); m.set_input_target(tgt)
ok, R = m.align(G)
print("gpu", m.last_gicp_stats, sm.se3_error(R, T))
ds, dt = m.get_downsampled(0), m.get_downsampled(1)
t = time.time()
o = ong.ndt_gicp_align(src, tgt, G, downsampled=(ds, dt))
print("oracle %.1fs" % (time.time() - t), o["ndt"]["iterations"], o["ndt"]["score"], o["gicp"]["iterations"], o["gicp"]["score"], sm.se3_error(o["result"], T))
print("gpu vs oracle", sm.se3_error(R, o["result"]))
print("gpu vs ndt-oracle", sm.se3_error(R, o["ndt"]["result"]), "oracle gicp vs its ndt", sm.se3_error(o["result"], o["ndt"]["result"]))
tr = []
g = ong.gicp_align(ds, dt, o["ndt"]["result"].astype(np.float32), trace=tr)
for r in tr: print(r)
