import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
a, b, T = synth.scan_pair("cfg2", n_points=120000)
m = sm.IcpFastHip(pair_slots=2, max_source_points=120000, max_target_points=120000 // 4 + 64)
m.prepare_target(a)
t = time.time()
for _ in range(10): M = m.prepare_target(a)
print("prepare_target (upload + device CalculateNormals): %.2f ms" % ((time.time() - t) / 10 * 1e3), M)
m.set_input_source(a, slot=1)
t = time.time()
for _ in range(10): M = m.prepare_target_from_source(1, 0)
print("prepare_target_from_source (device only): %.2f ms" % ((time.time() - t) / 10 * 1e3), M)
t = time.time(); q, n = sm.calculate_normals(a[:, :3].astype(np.float64)); print("host CalculateNormals: %.1f ms" % ((time.time() - t) * 1e3))
t = time.time()
for _ in range(10): m.set_input_source(a, slot=1)
print("set_input_source (host convert + Morton sort + upload): %.2f ms" % ((time.time() - t) / 10 * 1e3))
