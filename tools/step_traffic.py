"""HBM traffic of ONE 512-pair, 20-iteration batch summed over every launch, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
csv output) of a run that aligns the same batch `steps` times (tools/fused_probe.py pairs=512 steps=2 cfg="fused:" = 5 identical
batches: 2 warm-up, 2 timed, 1 profiled).  Per kernel: launches per step, bytes per step = (fetch_factor x FETCH_SIZE + write_factor x
WRITE_SIZE) x 1024 summed over its launches / steps; the total is the step's traffic.
Usage: step_traffic.py <fetch dir> <write dir> [calibration.json] [steps in the run] [workload text]"""
import csv, glob, json, os, sys, collections


def sums(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smhip::", "")
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


fetch, write = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
ff, wf, src = 2.0, 1.0, "MI355X_MICROARCH.md: FETCH_SIZE x 2, WRITE_SIZE as reported"
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    c = json.load(open(sys.argv[3]))
    if c.get("fetch_factor_stream_12_4_4") and c.get("write_factor_dword"):
        ff, wf, src = float(c["fetch_factor_stream_12_4_4"]), float(c["write_factor_dword"]), os.path.basename(sys.argv[3]) + " (streaming shapes)"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
workload = sys.argv[5] if len(sys.argv) > 5 else "tools/profile_target.py B=512 reps=1: 512 copies of the cfg2 120 k-point pair, guess 0.6 m off, 20 iterations, two 256-pair halves on two streams"
rows = {}
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("nn_", "listed_plan", "accumulate", "iteration_sums", "finalize", "grid_", "tgt_reduce", "reset_scratch", "pack_source", "pose_setup")):
        continue
    nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
    rows[k] = {"launches": max(nf, nw) / steps, "fetch_kb": round(f / steps), "write_kb": round(w / steps), "bytes": int((ff * f + wf * w) * 1024 / steps)}
total = sum(r["bytes"] for r in rows.values())
print(json.dumps({"workload": workload, "steps_in_the_run": steps,
                  "fetch_factor": ff, "write_factor": wf, "factors_from": src,
                  "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["bytes"])),
                  "step_total_bytes": total, "step_total_GB": round(total / 1e9, 2),
                  "algorithmic_bytes_per_step": int(512 * 88.84e6), "ratio_to_algorithmic": round(total / (512 * 88.84e6), 3)}, indent=1))
