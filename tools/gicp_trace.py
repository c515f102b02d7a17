import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
from oracle import ndt_gicp as ong
a, b, T = synth.scan_pair("cfg2", n_points=12000)
src, tgt = b[:, :3].copy(), a[:, :3].copy()
G = synth.make_pose(t=(0.6, 0, 0))
ds, dt = ong.approximate_voxel_grid(src, 0.2), ong.approximate_voxel_grid(tgt, 0.2)
o = ong.ndt_gicp_align(src, tgt, G, downsampled=(ds, dt))
g0 = o["ndt"]["result"].astype(np.float32)
m = sm.NdtGicpHip(max_source_points=16384, max_target_points=16384)
for k in (1,):
    m.set_gicp_options(gicp_max_iterations=k)
    fit, R = m.gicp_only(ds, dt, g0.astype(np.float64))
    tr = []
    w = ong.gicp_align(ds, dt, g0, max_iterations=k, trace=tr, debug=True)
    print(k, "gpu it", m.last_gicp_stats["gicp_iterations"], "evals", m.last_gicp_stats["gicp_function_evaluations"], "oracle it", w["iterations"],
          "diff", sm.se3_error(R, w["result"].astype(np.float64)), "delta", [round(t["delta"], 3) for t in tr], "inner", [t["inner"] for t in tr])
cg = m.get_covariances(1, len(dt)); cw = ong.gicp_covariances(dt)
print("cov err quantiles", np.quantile(np.abs(cg - cw).max(axis=(1, 2)), [0.5, 0.9, 0.99, 1.0]))
