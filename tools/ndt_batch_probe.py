"""Ndt::Align at BASELINE config #3 (120 k-point scan vs 500 k-point submap, 1 m voxels): single calls against the lock-step batch
(smhip_ndt_align_batch), everything rebuilt per Align and with the voxel tables kept.  usage: ndt_batch_probe.py [K=64] [distinct=8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import staticmapping_amd as sm
import bench

kv = dict(a.split("=") for a in sys.argv[1:])
K = int(kv.get("K", 64)); D = int(kv.get("distinct", 8))
dev = torch.device("cuda", 0)
cases = []
for k in range(D):
    src, tgt, T, G = bench._submap_case(5, 500_000, seed=31 + k, device=dev)
    cases.append(dict(src=src, tgt=tgt, T=T, guess=G))
ns = max(len(c["src"]) for c in cases); nt = max(len(c["tgt"]) for c in cases)
m1 = sm.NdtHip(max_source_points=ns, max_target_points=nt)
single = []
t_single = 0.0
for c in cases:
    m1.set_input_source(c["src"]); m1.set_input_target(c["tgt"])
    m1.set_target_cache(False)
    m1.align(c["guess"])
    t0 = time.perf_counter(); ok, R = m1.align(c["guess"]); t_single += time.perf_counter() - t0
    single.append((R, m1.get_fitness_score(), dict(m1.last_ndt_stats)))
m1.close()
print(f"single, rebuilt: {t_single / D * 1e3:.2f} ms per Align = {D / t_single:.0f}/s; iterations {[s[2]['iterations'] for s in single]}", flush=True)
mb = sm.NdtHip(max_source_points=ns, max_target_points=nt, pair_slots=K)
for k in range(K):
    c = cases[k % D]
    mb.set_input_source(c["src"], slot=k); mb.set_input_target(c["tgt"], slot=k)
g = [cases[k % D]["guess"] for k in range(K)]
for cache in (False, True):
    mb.set_target_cache(cache)
    mb.align_batch(K, g)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        R, sc, st = mb.align_batch(K, g)
    dt = (time.perf_counter() - t0) / reps
    same = all(R[k].tobytes() == single[k % D][0].tobytes() and sc[k] == single[k % D][1] for k in range(K))
    print(f"batch of {K}, tables {'kept' if cache else 'rebuilt'}: {dt * 1e3:.2f} ms per batch = {K / dt:.0f} Aligns/s; equal to the single calls bit for bit: {same}", flush=True)
mb.close()
