"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output), written as the
json bench.py reads (profiles/traffic_<kernel>.json).  The counters are turned into bytes with the factors calibrated on
known-byte kernels of the same access shape (tools/traffic_calib.sh -> profiles/r03_traffic_calibration.json): ONE number
per kernel instead of the bracket the guide's 16 B/lane rule left (MI355X_MICROARCH.md, HBM section).
Usage: traffic_json.py <fetch dir> <write dir> <kernel substring> <pairs per launch> <source points> <algorithmic B/pt>
                       <implementation B/pt> [calibration.json] [json name of the kernel]"""
import csv, glob, hashlib, json, os, sys


def avg(d, name, counter):
    v = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)


fetch_dir, write_dir, kernel, pairs, ns, alg, impl = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])
calib_path = sys.argv[8] if len(sys.argv) > 8 else ""
fetch, nf = avg(fetch_dir, kernel, "FETCH_SIZE")
write, nw = avg(write_dir, kernel, "WRITE_SIZE")
ff, wf, src = 2.0, 1.0, "MI355X_MICROARCH.md: FETCH_SIZE x 2 (16 B/lane streaming reads), WRITE_SIZE as reported (uncalibrated)"
if calib_path and os.path.exists(calib_path):
    c = json.load(open(calib_path))
    if c.get("fetch_factor_stream_12_4_4") and c.get("write_factor_dword"):
        ff, wf = float(c["fetch_factor_stream_12_4_4"]), float(c["write_factor_dword"])
        src = (f"{os.path.basename(calib_path)}: true bytes / (counter x 1024) of a known-byte kernel reading a 12-byte row + int + float and "
               f"writing one float per element (the shape of nn_certify / accumulate): FETCH_SIZE x {ff:.4f}, WRITE_SIZE x {wf:.4f}")
# the record is dated with the kernel sources it was measured on: bench.py uses it only while they are unchanged
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["staticmapping_amd/csrc/icp_kernels.hip", "staticmapping_amd/csrc/smhip_device.h"]
_h = hashlib.sha256()
for _f in sorted(SOURCES):
    with open(os.path.join(ROOT, _f), "rb") as _fh:
        _h.update(_fh.read())
print(json.dumps({
    "kernel": sys.argv[9] if len(sys.argv) > 9 else kernel, "source_files": SOURCES, "source_sha": _h.hexdigest(), "pairs_per_launch": pairs, "nn_mode": "grid", "source_points": ns,
    "fetch_size_kb_per_launch": round(fetch), "write_size_kb_per_launch": round(write), "launches_averaged": [nf, nw],
    "fetch_factor": ff, "write_factor": wf, "factors_from": src,
    "hbm_bytes_per_launch": int((ff * fetch + wf * write) * 1024),
    "hbm_bytes_per_launch_uncorrected": int((fetch + write) * 1024),
    "algorithmic_bytes_per_launch": int(alg * pairs * ns),
    "compulsory_bytes_of_this_implementation": int(impl * pairs * ns),
    "ratio_to_algorithmic": round((ff * fetch + wf * write) * 1024 / max(1.0, alg * pairs * ns), 4),
    "how": "tools/r06_final.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and a separate --pmc WRITE_SIZE pass of "
           "tools/fused_probe.py pairs=512 distinct=512 (the bench workload, two 256-pair halves on two streams), averaged over the kernel's launches; "
           "bytes = (fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE) * 1024 (the counters are in KB)",
}, indent=1))
