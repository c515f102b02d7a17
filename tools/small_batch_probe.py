"""A handful of 120 k-point IcpFast pairs in one call (the back end's concurrent submap pairs, builder/map_builder.cc:399-446): the one
cooperative launch (a row of its grid per pair) against the separate launches.  usage: small_batch_probe.py [pairs=1,2,4,6,8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
kv = dict(x.split("=") for x in sys.argv[1:])
a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
guess = T @ synth.make_pose(rpy_deg=(0.0, 0.0, 0.2), t=(0.03, 0.01, 0.0))
for P in [int(x) for x in kv.get("pairs", "1,2,4,6,8").split(",")]:
    for sep in (0, 1):
        m = sm.IcpFastHip(pair_slots=P, max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0, no_single_kernel=sep)
        for s in range(P):
            m.set_input_source(b, slot=s); m.set_input_target(q, n, slot=s)
        m.align_batch(P, [guess] * P)
        t = time.time(); reps = 10
        for _ in range(reps): m.align_batch(P, [guess] * P)
        dt = (time.time() - t) / reps
        print(f"{P} pairs, {'separate launches' if sep else 'one launch'}: {dt * 1e3:.3f} ms per batch = {P / dt:.0f} alignments/s")
        m.close()
