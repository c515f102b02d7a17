"""Experiment: two 256-pair handles on their own streams, the second started half an alignment after the first, so that one
handle's search-heavy early iterations overlap the other's streaming late iterations -- against one 512-pair handle whose
two halves run in lockstep.  usage: phase_probe.py [shift_ms ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
from oracle import cref

a, b, T = synth.scan_pair("cfg2", n_points=120000)
q, n, _ = cref.calculate_normals(a[:, :3].astype(np.float64))
ok = np.isfinite(n).all(axis=1); q, n = q[ok], n[ok]
guess = synth.make_pose(t=(0.6, 0, 0))


def make(B, **kw):
    m = sm.IcpFastHip(pair_slots=B, max_source_points=len(b), max_target_points=len(q), max_iteration=20, early_exit=0, **kw)
    m.set_input_source(b); m.set_input_target(q, n)
    for s in range(1, B): m.copy_slot(0, s)
    return m


m = make(512)
g = [guess] * 512
m.align_batch(512, g)
t = time.time()
for _ in range(4): m.enqueue_batch(512, g)
m.synchronize()
base = 4 * 512 / (time.time() - t)
print(f"one handle, 512 pairs, two lockstep halves: {base:.0f} align/s", flush=True)
m.close()

shifts = [float(x) for x in sys.argv[1:]] or [0.0, 6.0, 11.0, 16.0]
for streams in (1, 2):
    A = make(256, no_overlap=1 if streams == 1 else 0); Bm = make(256, no_overlap=1 if streams == 1 else 0)
    gg = [guess] * 256
    A.align_batch(256, gg); Bm.align_batch(256, gg)
    for shift in shifts:
        A.synchronize(); Bm.synchronize()
        reps = 6
        t = time.time()
        A.enqueue_batch(256, gg)
        time.sleep(shift * 1e-3)
        for _ in range(reps - 1):
            Bm.enqueue_batch(256, gg)       # blocks until this handle's previous batch is done: keeps the phase
            A.enqueue_batch(256, gg)
        Bm.enqueue_batch(256, gg)
        A.synchronize(); Bm.synchronize()
        dt = time.time() - t
        print(f"two handles x 256 pairs ({streams} stream(s) each), shift {shift:4.1f} ms: {2 * reps * 256 / dt:.0f} align/s", flush=True)
    R, sc, st = A.fetch_batch(256)
    print("  check", sm.se3_error(R[0], T))
    A.close(); Bm.close()
