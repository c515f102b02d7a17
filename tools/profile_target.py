"""Workload for rocprofv3: B pairs, `reps` batched alignments of BASELINE config #2."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
kv = dict(a.split("=") for a in sys.argv[1:])
B = int(kv.get("B", 64)); reps = int(kv.get("reps", 2)); cell = float(kv.get("cell", 0.25)); ring = int(kv.get("ring", 8))
mode = int(kv.get("mode", 1)); n_points = int(kv.get("n", 120000)); noov = int(kv.get("noov", 0)); nocert = int(kv.get("nocert", 0)); nolds = int(kv.get("nolds", 0))
a, b, T = synth.scan_pair("cfg2", n_points=n_points)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
guess = synth.make_pose(t=(0.6, 0, 0))
m = sm.IcpFastHip(pair_slots=B, max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0,
                  nn_mode=mode, grid_cell=cell, grid_max_ring=ring, no_overlap=noov, no_certify=nocert, no_lds_table=nolds)
m.set_input_source(b); m.set_input_target(q, n)
for s in range(1, B): m.copy_slot(0, s)
for _ in range(reps): R, sc, st = m.align_batch(B, [guess] * B)
print("ok", sm.se3_error(R[0], T), st[0])
