"""A/B probe of the batched IcpFast iteration on the bench workload (distinct consecutive pairs of the synthetic drive):
every configuration on the same handle and clouds, alignments/s, the per-class kernel times, how many iterations the fused
certificate pass carried, and the agreement of the poses with the first configuration.
usage: fused_probe.py [pairs=512] [distinct=64] [steps=6] [guess=cv|id|mix] [cfg=name:opt=val,opt=val;name:...] [env=NAME=V,NAME=V]
  e.g. cfg="separate:no_fused_sums=1;fused:" """
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import staticmapping_amd as sm  # noqa: E402
import bench  # noqa: E402

kv = dict(a.split("=", 1) for a in sys.argv[1:])
B = int(kv.get("pairs", 512)); D = int(kv.get("distinct", 64)); steps = int(kv.get("steps", 6)); gk = kv.get("guess", "cv")
cfgs = []
for part in kv.get("cfg", "separate:no_fused_sums=1;fused:").split(";"):
    name, _, rest = part.partition(":")
    opts, env = {}, {}
    for item in filter(None, rest.split(",")):
        k, v = item.split("=")
        if k.isupper():
            env[k] = v
        else:
            opts[k] = float(v) if "." in v else int(v)
    cfgs.append((name, opts, env))

dev = torch.device("cuda", 0)
work = bench.build_workload(D, bench.N_POINTS, dev)
ns = max(len(w["src"]) for w in work); nt = max(len(w["q"]) for w in work)
m = sm.IcpFastHip(pair_slots=B, max_source_points=ns, max_target_points=nt, max_iteration=20, early_exit=0)
for s in range(B):
    w = work[s % D]
    m.set_input_source(w["src"], slot=s); m.set_input_target(w["q"], w["n"], slot=s)
m.synchronize()
if gk == "mix":
    guesses = [work[s % D]["guess_cv" if s % 2 == 0 else "guess_id"] for s in range(B)]
else:
    guesses = [work[s % D]["guess_cv" if gk == "cv" else "guess_id"] for s in range(B)]
base_opts = dict(no_fused_sums=0, split_after=0, overlap_streams=0, no_overlap=0)
ref = None
for name, opts, env in cfgs:
    for k in ("SMHIP_BAND_PAD", "SMHIP_BAND_GAIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m.set_options(**{**base_opts, **opts})
    for _ in range(2):
        m.enqueue_batch(B, guesses); R, sc, st = m.fetch_batch(B)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.enqueue_batch(B, guesses)
    R, sc, st = m.fetch_batch(B)
    dt = (time.perf_counter() - t0) / steps
    m.enable_profile(True); m.enqueue_batch(B, guesses); m.fetch_batch(B); p = m.get_profile(); m.enable_profile(False)
    fused = np.array([s["fused_iterations"] for s in st]); searched = np.mean([s["searched_queries"] for s in st])
    line = (f"[{name}] {B / dt:9.1f} align/s  {dt * 1e3:7.2f} ms/step  split={p['split_after_used']} fused_iters mean {fused.mean():.1f} min {fused.min()} max {fused.max()} "
            f"searched {searched:.0f}  ms: main {p['ms_nn_main']:.2f} certify {p['ms_nn_certify']:.2f} listed {p['ms_nn_listed']:.2f} refine {p['ms_nn_refine']:.2f} "
            f"acc {p['ms_error_elements']:.2f} solve {p['ms_solve']:.2f} prep {p['ms_prepare']:.2f}")
    if ref is None:
        ref = (R, [s["kept"] for s in st], [s["limit_d2"] for s in st])
    else:
        e = [sm.se3_error(R[s], ref[0][s]) for s in range(B)]
        kept_same = sum(int(st[s]["kept"] == ref[1][s]) for s in range(B)); lim_same = sum(int(st[s]["limit_d2"] == ref[2][s]) for s in range(B))
        line += f"  vs first: rot {max(x[0] for x in e):.2e} trans {max(x[1] for x in e):.2e} kept== {kept_same}/{B} limit== {lim_same}/{B}"
    print(line, flush=True)
errs = [sm.se3_error(R[s], work[s % D]["T"]) for s in range(B)]
print("last cfg vs truth: median trans", float(np.median([e[1] for e in errs])), flush=True)
m.close()
