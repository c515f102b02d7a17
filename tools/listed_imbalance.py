"""How unevenly the queries whose certificate failed are spread over the pairs of a batch, per iteration (bench workload).
usage: listed_imbalance.py [guess=cv|id|mix]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import staticmapping_amd as sm
import bench
kv = dict(a.split("=", 1) for a in sys.argv[1:])
gk = kv.get("guess", "cv"); B = 512
work = bench.build_workload(B, bench.N_POINTS, torch.device("cuda", 0))
ns = max(len(w["src"]) for w in work); nt = max(len(w["q"]) for w in work)
m = sm.IcpFastHip(pair_slots=B, max_source_points=ns, max_target_points=nt, max_iteration=20, early_exit=0)
for s in range(B):
    m.set_input_source(work[s]["src"], slot=s); m.set_input_target(work[s]["q"], work[s]["n"], slot=s)
g = [work[s]["guess_cv" if (gk == "cv" or (gk == "mix" and s % 2 == 0)) else "guess_id"] for s in range(B)]
for _ in range(3):
    m.align_batch(B, g)
c = np.array([m.search_counts(s) for s in range(B)])
print("split", m.get_profile()["split_after_used"])
for it in range(c.shape[1]):
    v = c[:, it]
    print(f"iteration {it:2d}: searched per pair min {v.min():6d} median {int(np.median(v)):6d} mean {v.mean():8.0f} p90 {int(np.percentile(v, 90)):6d} max {v.max():6d}  max/mean {v.max() / max(1.0, v.mean()):.2f}  passes of 8192: max {int(np.ceil(v.max() / 8192))} mean {np.mean(np.ceil(v / 8192)):.2f}")
m.close()
