"""Queries that needed a search, per iteration (difference of the cumulative counter over runs of k iterations)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
guess = synth.make_pose(t=(0.6, 0, 0))
m = sm.IcpFastHip(pair_slots=1, max_source_points=len(b), max_target_points=len(q), early_exit=0)
m.set_input_source(b); m.set_input_target(q, n)
prev = (0, 0)
for k in range(1, 21):
    m.set_options(max_iteration=k)
    m.align(guess)
    st = m.last_stats[0]
    print(k - 1, "searched", st["searched_queries"] - prev[0], "hard", st["hard_queries"] - prev[1], "limit", round(float(np.sqrt(st["limit_d2"])), 4))
    prev = (st["searched_queries"], st["hard_queries"])
