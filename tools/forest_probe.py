"""CalculateNormals of a batch of scans on the device (smhip_prepare_targets_from_sources: a forest of kd-trees, one workgroup per scan)
for several batch sizes: ms per call and per scan.  usage: forest_probe.py [scans=64,128,256,384,512] [n=120000]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
kv = dict(x.split("=") for x in sys.argv[1:])
n = int(kv.get("n", 120000))
a, b, T = synth.scan_pair("cfg2", n_points=n)
rng = np.random.default_rng(3)
def rows4(c):
    out = np.zeros((len(c), 4), np.float32); out[:, :3] = c[:, :3]; return out
base = [rows4(a), rows4(b)]
for S in [int(x) for x in kv.get("scans", "64,128,256,384,512").split(",")]:
    m = sm.IcpFastHip(pair_slots=2 * S, max_source_points=n, max_target_points=n // 4 + 64)
    scans = []
    for k in range(S):
        sc = base[k & 1].copy()
        sc[:, :3] += rng.normal(0, 1e-3, (len(sc), 3)).astype(np.float32)
        scans.append(sc)
    for k0 in range(0, S, 128):
        m.set_input_sources_batch(scans[k0:k0 + 128], list(range(k0, min(S, k0 + 128))))
    fr, to = list(range(S)), list(range(S, 2 * S))
    m.prepare_targets_from_sources(fr, to); m.synchronize()
    t = time.perf_counter(); reps = 5
    for _ in range(reps): Ms = m.prepare_targets_from_sources(fr, to)
    dt = (time.perf_counter() - t) / reps
    print(f"{S} scans of {n} points: {dt * 1e3:.2f} ms per call = {dt * 1e6 / S:.1f} us per scan (targets of {int(np.mean(Ms))} points)", flush=True)
    m.close()
