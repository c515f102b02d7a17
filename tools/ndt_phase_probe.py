"""Where a workgroup of ndt_derivatives_ctl spends its time (tuning build: SMHIP_EXTRA_HIPCC_FLAGS=-DNDT_TIMING): per-phase stamps
of every workgroup of the last evaluation of a config #3 Align.  usage: ndt_phase_probe.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import staticmapping_amd as sm
from staticmapping_amd import _capi
import bench
dev = torch.device("cuda", 0)
src, tgt, T, G = bench._submap_case(5, 500_000, 4, dev)
m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
m.set_input_source(src); m.set_input_target(tgt)
m.align(G); m.align(G)
lib = _capi.load_library()
f = lib.smhip_ndt_debug_rows
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
rows = ctypes.c_int(0)
buf = np.zeros((4096, 32), dtype=np.uint64)
assert f(m._h, buf.ctypes.data, 4096, ctypes.byref(rows)) == 0
w = buf[:rows.value]
t0 = w[:, 29].astype(np.int64); pk = w[:, 30]; t5 = (w[:, 31] & np.uint64(0xffffffff)).astype(np.int64); cyc = (w[:, 31] >> np.uint64(32)).astype(np.int64)
d = [((pk >> np.uint64(16 * k)) & np.uint64(0xffff)).astype(np.int64) for k in range(4)] + [t5]
names = ["src+transform+records", "18 words + count", "scan + emit", "phase B", "epilogue"]
prev = np.zeros_like(t5)
print(f"workgroups {rows.value}; ticks of 10 ns")
for n_, x in zip(names, d):
    print(f"  {n_:24s} mean {np.mean(x - prev) / 100:7.2f} us   max {np.max(x - prev) / 100:7.2f} us")
    prev = x
print(f"  workgroup lifetime       mean {np.mean(t5) / 100:7.2f} us   max {np.max(t5) / 100:7.2f} us")
print(f"  first start .. last end  {(np.max(t0 + t5) - np.min(t0)) / 100:7.2f} us; starts spread over {(np.max(t0) - np.min(t0)) / 100:7.2f} us")
print(f"  shader clock while the workgroups ran: {np.mean(cyc / np.maximum(t5, 1)) * 100:.0f} MHz (cycle counter / 100 MHz real-time counter)")
m.close()
