"""Diagnostic (needs a library built with -DSMHIP_PHASE_TIMING=1): per-phase cycle shares of nn_ball_lds (SMHIP_DEBUG_FLAGS=16), 64 pairs, 8 iterations, one stream."""
import os, sys
os.environ["SMHIP_DEBUG_FLAGS"] = "16"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
B = 64
guess = synth.make_pose(t=(0.6, 0, 0))
m = sm.IcpFastHip(pair_slots=B, max_source_points=len(b), max_target_points=len(q), max_iteration=8, early_exit=0, no_overlap=1)
m.set_input_source(b); m.set_input_target(q, n)
for s in range(1, B): m.copy_slot(0, s)
m.align_batch(B, [guess] * B)
m.close()
