"""Race hunt for the cooperative single-pair launch: small clouds on a grid of mostly idle workgroups (SMHIP_ONE_BLOCKS + SMHIP_ONE_IDLE),
many repeated Aligns, every one compared with the first bit for bit.  usage: one_race_probe.py [n=5000] [reps=200] [pairs=1] [early=1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
kv = dict(x.split("=") for x in sys.argv[1:])
n, reps, P, early = int(kv.get("n", 5000)), int(kv.get("reps", 200)), int(kv.get("pairs", 1)), int(kv.get("early", 1))
tgt, src, T = synth.three_planes_pair(5000, seed=1)
q, nr = sm.calculate_normals(tgt[:, :3].astype(np.float64))
a, b, T2 = synth.scan_pair("cfg2", n_points=20000)
q2, n2 = sm.calculate_normals(a[:, :3].astype(np.float64))
clouds = [(src[:n], q, nr, np.eye(4)), (b, q2, n2, synth.make_pose(t=(0.6, 0, 0)))]
m = sm.IcpFastHip(pair_slots=P, max_source_points=20000, max_target_points=max(len(q), len(q2)), max_iteration=25, early_exit=early)
mix = kv.get("mix", "ab")              # which cloud each slot holds: a = the three-planes cloud, b = the 20 k scan
pick = [0 if mix[s % len(mix)] == "a" else 1 for s in range(P)]
for s in range(P):
    c = clouds[pick[s]]
    m.set_input_source(c[0], slot=s); m.set_input_target(c[1], c[2], slot=s)
g = [clouds[pick[s]][3] for s in range(P)]
R0, s0, st0 = m.align_batch(P, g)
bad = 0
t = time.time()
for r in range(reps):
    R, sc, st = m.align_batch(P, g)
    if R.tobytes() != R0.tobytes() or sc.tobytes() != s0.tobytes():
        bad += 1
        if bad <= 3: print("run", r, "differs:", [ (x["iterations"], x["kept"], x["hard_queries"], x["searched_queries"]) for x in st], "first:", [(x["iterations"], x["kept"], x["hard_queries"], x["searched_queries"]) for x in st0])
dt = time.time() - t
used, fell = m.single_launch_counts()
print(f"n={n} pairs={P} reps={reps}: {bad} runs differ from the first; {dt / reps * 1e3:.3f} ms per call; {used} of {reps + 1} calls ran as one launch, {fell} stopped themselves")
