"""NdtWithGicp at BASELINE config #5 (a 120 k-point scan against a 2 M-point submap): single calls against the lock-step batch
(smhip_ndt_gicp_align_batch), rebuilt and with the targets kept, for one or more search cells.
usage: gicp_batch_probe.py [jobs=16] [reps=3] [cells=0.6,0.5] [single=1] [mode=both|rebuilt|kept]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import staticmapping_amd as sm  # noqa: E402
from staticmapping_amd import synth  # noqa: E402
import bench  # noqa: E402

kv = dict(a.split("=", 1) for a in sys.argv[1:])
J = int(kv.get("jobs", 16)); reps = int(kv.get("reps", 3)); mode = kv.get("mode", "both")
cells = [float(c) for c in kv.get("cells", "0.6").split(",")]
dev = torch.device("cuda", 0)
src, tgt, T, G = bench._submap_case(20, 2_000_000, 6, dev)
gs = [G @ synth.make_pose(t=(0.01 * (k % 8), -0.01 * (k % 5), 0.0), rpy_deg=(0, 0, 0.05 * (k % 7))) for k in range(J)]
gs[0] = G
modes = [("rebuilt", False), ("kept", True)] if mode == "both" else [(mode, mode == "kept")]
R1 = None
for cell in cells:
    if int(kv.get("single", 1)):
        m1 = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt), gicp_search_cell=cell)
        m1.set_input_source(src); m1.set_input_target(tgt)
        for name, cache in modes:
            m1.set_target_cache(cache)
            m1.align(G)
            t0 = time.perf_counter()
            for _ in range(5):
                ok, R1 = m1.align(G)
            dt = (time.perf_counter() - t0) / 5
            print(f"cell {cell}: single {name}: {1 / dt:8.1f} /s ({1e3 * dt:.2f} ms)  stats {m1.last_gicp_stats}", flush=True)
        m1.close()
    mb = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt), jobs=J, gicp_search_cell=cell)
    for k in range(J):
        mb.set_input_source(src, slot=k); mb.set_input_target(tgt, slot=k)
    for name, cache in modes:
        mb.set_target_cache(cache)
        mb.align_batch(J, gs)
        t0 = time.perf_counter()
        for _ in range(reps):
            R, sc, st = mb.align_batch(J, gs)
        dt = (time.perf_counter() - t0) / reps
        same = R1 is not None and np.array_equal(R[0], R1)
        print(f"cell {cell}: batch of {J} {name}: {J / dt:8.1f} /s ({1e3 * dt / J:.2f} ms per pair)  job 0 equal to the single call: {same}  "
              f"evals {[s['gicp_function_evaluations'] for s in st][:8]} iters {[s['gicp_iterations'] for s in st][:8]}", flush=True)
    mb.close()
