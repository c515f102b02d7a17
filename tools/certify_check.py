"""Certificates must not change a single bit: a 120 k-point Align with certificates (fused kernel, and the two-launch form)
against the same Align with every query searched in every iteration (no_certify) and against the plain ring search."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n = sm.calculate_normals(a[:, :3].astype(np.float64))
for gname, guess in (("offset 0.6 m", synth.make_pose(t=(0.6, 0, 0))), ("identity", np.eye(4)), ("truth", T)):
    ref = None
    for name, opts in (("no_certify", dict(no_certify=1)), ("fused", dict()), ("split 1", dict(split_after=1)), ("split 3", dict(split_after=3)),
                       ("global variant", dict(no_lds_table=1)), ("ring", dict(use_ball=0))):
        m = sm.IcpFastHip(max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0, **opts)
        m.set_input_source(b); m.set_input_target(q, n)
        ok, R = m.align(guess)
        st = m.last_stats[0]
        m.close()
        key = (R.tobytes(), st["kept"], st["limit_d2"])
        if ref is None:
            ref = key
        print(f"{gname:14s} {name:15s} identical={key == ref} kept={st['kept']} limit={st['limit_d2']:.9g} searched={st['searched_queries']} hard={st['hard_queries']} "
              f"d={sm.se3_error(R, np.frombuffer(ref[0]).reshape(4, 4))}")
