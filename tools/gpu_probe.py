"""Quick GPU timing probe (not the bench): batched ICP with the per-kernel profile.
usage: gpu_probe.py NPOINTS B1,B2 [key=value ...]   keys: cell ring morton mode"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
from oracle import cref

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
batches = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "16", "64"])]
kv = dict(a.split("=") for a in sys.argv[3:])
cells = [float(x) for x in kv.get("cell", "0.25").split(",")]
rings = [int(x) for x in kv.get("ring", "8").split(",")]
mortons = [int(x) for x in kv.get("morton", "0").split(",")]
modes = [int(x) for x in kv.get("mode", "1").split(",")]
tiles = [int(x) for x in kv.get("ball", "1").split(",")]
margins = [float(x) for x in kv.get("radius", "0.3").split(",")]
capf = float(kv.get("capf", "1.5")); twop = int(kv.get("nocert", "0")); nolds = int(kv.get("nolds", "0")); noov = int(kv.get("noov", "0")); nstreams = int(kv.get("streams", "0")); split = int(kv.get("split", "0"))


def morton_order(p, cell=0.25):
    c = np.floor((p - p.min(0)) / cell).astype(np.uint64)
    def spread(v):
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    code = spread(c[:, 0]) | (spread(c[:, 1]) << np.uint64(1)) | (spread(c[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


a, b0, T = synth.scan_pair("cfg2", n_points=n_points)
q, n, _ = cref.calculate_normals(a[:, :3].astype(np.float64))
ok = np.isfinite(n).all(axis=1); q, n = q[ok], n[ok]
guess = synth.make_pose(t=(0.6, 0, 0))
print("ns", len(b0), "nt", len(q), flush=True)
for mode in modes:
  for morton in mortons:
    b = b0[morton_order(b0[:, :3].astype(np.float64))] if morton else b0
    for cell in cells:
      for ring in rings:
       for tile in tiles:
        for margin in margins:
         for B in batches:
            m = sm.IcpFastHip(pair_slots=B, max_source_points=len(b), max_target_points=len(q), use_ball=tile, ball_radius=margin, ball_cap_factor=capf, no_certify=twop, no_lds_table=nolds, no_overlap=noov, overlap_streams=nstreams, split_after=split,
                              max_iteration=20, early_exit=0, nn_mode=mode, grid_cell=cell, grid_max_ring=ring)
            m.set_input_source(b); m.set_input_target(q, n)
            for s in range(1, B): m.copy_slot(0, s)
            g = [guess] * B
            m.align_batch(B, g)  # warm
            t = time.time(); reps = 3
            for _ in range(reps): R, sc, st = m.align_batch(B, g)
            dt = (time.time() - t) / reps
            m.enable_profile(True); m.align_batch(B, g); p = m.get_profile(); m.enable_profile(False)
            print(f"mode={ {0: 'brute', 1: 'grid', 2: 'nabo'}[mode] } ball={tile} radius={margin} cell={cell} ring={ring} B={B} {dt*1e3:.2f} ms/batch "
                  f"{B/dt:.1f} align/s hard={st[0]['hard_queries']} refined={st[0]['refined_iterations']} searched={st[0]['searched_queries']} fallback={st[0]['fallback_queries']} err={sm.se3_error(R[0], T)}", flush=True)
            print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()}, flush=True)
            m.close()
