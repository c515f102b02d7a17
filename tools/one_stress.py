"""Six host threads, a matcher each, single 120 k-point Aligns at the same time (the reference's back end runs six SubmapPairMatch
tasks at once, builder/map_builder.cc:399-446, 655): the cooperative single-pair launches of several handles must neither deadlock
(six grids of 256 resident workgroups do not fit the device together) nor disturb each other's results.  usage: one_stress.py [threads=6] [reps=10] [separate=1]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
kv = dict(x.split("=") for x in sys.argv[1:])
T_, reps, sep = int(kv.get("threads", 6)), int(kv.get("reps", 10)), int(kv.get("separate", 0))
a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
guess = T @ synth.make_pose(rpy_deg=(0.0, 0.0, 0.2), t=(0.03, 0.01, 0.0))
m = sm.IcpFastHip(max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0, no_single_kernel=sep)
m.set_input_source(b); m.set_input_target(q, n)
ref = m.align(guess)[1]; m.close()
out, err, counts = [None] * T_, [], [None] * T_
def work(k):
    try:
        m = sm.IcpFastHip(max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0, no_single_kernel=sep)
        m.set_input_source(b); m.set_input_target(q, n)
        for _ in range(reps):
            R = m.align(guess)[1]
            if not np.array_equal(R, ref):
                err.append((k, "differs"))
        out[k] = R
        counts[k] = m.single_launch_counts()
        m.close()
    except Exception as e:   # noqa: BLE001
        err.append((k, repr(e)))
t0 = time.time()
th = [threading.Thread(target=work, args=(k,)) for k in range(T_)]
for t in th: t.start()
for t in th: t.join()
dt = time.time() - t0
print(f"{T_} threads x {reps} Aligns: {dt * 1e3:.1f} ms wall = {T_ * reps / dt:.0f} Aligns/s; errors: {err}; (ran as one launch, stopped themselves) per matcher: {counts}")
