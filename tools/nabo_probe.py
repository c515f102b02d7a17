"""The bench's reference_search_eps3.16 figure alone, with per-iteration walked-query counts and the kernel-category
profile.  usage: nabo_probe.py [pairs=512] [distinct=64] [nocert=0,1] [guess=cv|id]"""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import staticmapping_amd as sm
import bench

kv = dict(a.split("=") for a in sys.argv[1:])
B = int(kv.get("pairs", "512")); D = int(kv.get("distinct", "64"))
nocerts = [int(x) for x in kv.get("nocert", "0").split(",")]
gkey = "guess_id" if kv.get("guess", "cv") == "id" else "guess_cv"
streams = int(kv.get("streams", "0"))
dev = torch.device("cuda", 0)
work = bench.build_workload(D, 120_000, dev)
ns = max(len(w["src"]) for w in work); nt = max(len(w["q"]) for w in work)
print(f"source points {ns}, target points {nt}", flush=True)
for nocert in nocerts:
    m = sm.IcpFastHip(pair_slots=B, max_source_points=ns, max_target_points=nt, max_iteration=20, early_exit=0,
                      nn_mode=sm.NN_NABO, nn_epsilon=3.16, no_certify=nocert, overlap_streams=streams)
    for s in range(B):
        m.set_input_source(work[s % D]["src"], slot=s); m.set_input_target(work[s % D]["q"], work[s % D]["n"], slot=s)
    g = [work[s % D][gkey] for s in range(B)]
    m.align_batch(B, g)
    t = time.perf_counter(); reps = 2
    for _ in range(reps):
        R, sc, st = m.align_batch(B, g)
    dt = (time.perf_counter() - t) / reps
    m.enable_profile(True); m.align_batch(B, g); p = m.get_profile(); m.enable_profile(False)
    print(f"nabo no_certify={nocert} guess={gkey} B={B} streams={streams} listed_blocks={os.environ.get('SMHIP_NABO_LISTED_BLOCKS', 'default')}: {dt * 1e3:.2f} ms/batch = {B / dt:.0f} align/s; walked per alignment "
          f"{np.mean([s['searched_queries'] for s in st]):.0f}; per iteration (slot 0) {m.search_counts(0)}", flush=True)
    print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()}, flush=True)
    m.close()
