# HBM traffic of ndt_derivatives_ctl per launch (config #3, K evaluations per launch): separate FETCH_SIZE / WRITE_SIZE passes,
# written as profiles/traffic_ndt_derivatives_ctl.json dated with the sha256 of the kernel's source.  usage: ndt_traffic.sh <tag> [K]
set -u
export TMPDIR=/tmp
tag=${1:-r06}; K=${2:-64}
out=$PWD/gpurun_out/${tag}_ndt_traffic
mkdir -p "$out"
python tools/ndt_traffic_probe.py K=$K > "$out/probe.txt" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python tools/ndt_traffic_probe.py K=$K > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python tools/ndt_traffic_probe.py K=$K > /dev/null 2> "$out/write.err"
python - "$out" $K <<'PY'
import csv, glob, hashlib, json, os, sys
out, K = sys.argv[1], int(sys.argv[2])
def avg(d, counter):
    v = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ndt_derivatives_ctl" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r.get("Grid_Size", "0") or 0) >= 256 * 400 * K:
                v.append(float(r["Counter_Value"]))
    v = v[-10:]                      # the timed launches (the Align's own rounds carry fewer running jobs at the end)
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)
fetch, nf = avg(out + "/fetch", "FETCH_SIZE")
write, nw = avg(out + "/write", "WRITE_SIZE")
probe = dict(kv.split("=") for kv in open(out + "/probe.txt").read().split() if "=" in kv)
files = ["staticmapping_amd/csrc/ndt_kernels.hip"]
hsh = hashlib.sha256()
for f in files:
    hsh.update(open(f, "rb").read())
pairs, ns = float(probe["pairs_per_launch"]), int(probe["ns"])
alg = 12.0 * ns * K + 36.0 * pairs
rec = {"kernel": "ndt_derivatives_ctl", "units_per_launch": K, "unit": "evaluations (one pair's computeDerivatives call each)",
       "source_files": files, "source_sha": hsh.hexdigest(),
       "fetch_size_kb_per_launch": round(fetch), "write_size_kb_per_launch": round(write), "launches_averaged": [nf, nw],
       "fetch_factor": 1.0, "write_factor": 1.0,
       "factors_from": "scattered 16-64 B gathers (voxel records, occupancy words): the counters at factor 1, as the scatter calibration of round 4 found for this access shape (profiles/r04_traffic_calibration.json calib_scatter); the streamed source read (16 B/point) is 1.9 MB of the total",
       "hbm_bytes_per_launch": int((fetch + write) * 1024),
       "algorithmic_bytes_per_launch": int(alg), "ratio_to_algorithmic": round((fetch + write) * 1024 / alg, 4),
       "ms_per_launch_untraced": float(probe["ms_per_launch"]),
       "how": "tools/ndt_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and a separate --pmc WRITE_SIZE pass of tools/ndt_traffic_probe.py, averaged over the last ten launches of the kernel (the timed ones)"}
json.dump(rec, open(out + "/traffic_ndt_derivatives_ctl.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
rm -rf "$out/fetch" "$out/write"
