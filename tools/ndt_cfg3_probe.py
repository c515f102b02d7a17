"""Ndt::Align at BASELINE config #3 exactly as bench.py's other_workloads times it (120 k-point scan vs 500 k-point submap of 5
merged scans, 1 m voxels): single Align rebuilt / kept, optional lock-step batch.  usage: ndt_cfg3_probe.py [K=0] [reps=10] [distinct=8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import staticmapping_amd as sm
import bench

kv = dict(a.split("=") for a in sys.argv[1:])
K = int(kv.get("K", 0)); reps = int(kv.get("reps", 10)); D = int(kv.get("distinct", 8))
dev = torch.device("cuda", 0)
src, tgt, T, G = bench._submap_case(5, 500_000, 4, dev)
m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
m.set_input_source(src); m.set_input_target(tgt)
for cache in (False, True):
    m.set_target_cache(cache)
    m.align(G)
    t = time.perf_counter()
    for _ in range(reps):
        ok, R = m.align(G)
    dt = (time.perf_counter() - t) / reps
    print(f"single {'kept' if cache else 'rebuilt'}: {dt * 1e3:.3f} ms per Align = {1 / dt:.0f}/s stats {m.last_ndt_stats} "
          f"score {m.get_fitness_score():.6f} err {sm.se3_error(R, T)}", flush=True)
R1 = R
m.close()
if K:
    cases = [dict(src=src, tgt=tgt, guess=G)]
    for k in range(1, D):
        s_, t_, T_, G_ = bench._submap_case(5, 500_000, seed=31 + k, device=dev)
        cases.append(dict(src=s_, tgt=t_, guess=G_))
    ns = max(len(c["src"]) for c in cases); nt = max(len(c["tgt"]) for c in cases)
    mb = sm.NdtHip(max_source_points=ns, max_target_points=nt, pair_slots=K)
    for k in range(K):
        c = cases[k % D]
        mb.set_input_source(c["src"], slot=k); mb.set_input_target(c["tgt"], slot=k)
    g = [cases[k % D]["guess"] for k in range(K)]
    for cache in (False, True):
        mb.set_target_cache(cache)
        mb.align_batch(K, g)
        t0 = time.perf_counter()
        for _ in range(3):
            R, sc, st = mb.align_batch(K, g)
        dt = (time.perf_counter() - t0) / 3
        print(f"batch of {K}, tables {'kept' if cache else 'rebuilt'}: {dt * 1e3:.2f} ms per batch = {K / dt:.0f} Aligns/s; slot 0 equal to the single call: "
              f"{R[0].tobytes() == R1.tobytes()}; calls {[s['derivative_calls'] for s in st][:8]}", flush=True)
    mb.close()
