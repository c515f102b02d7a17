"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output), written as the
json bench.py reads (profiles/traffic_<kernel>.json).
Usage: traffic_json.py <fetch dir> <write dir> <kernel substring> <pairs per launch> <source points> <algorithmic B/pt> <implementation B/pt>"""
import csv, glob, json, sys


def avg(d, name, counter):
    v = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)


fetch_dir, write_dir, kernel, pairs, ns, alg, impl = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])
fetch, nf = avg(fetch_dir, kernel, "FETCH_SIZE")
write, nw = avg(write_dir, kernel, "WRITE_SIZE")
print(json.dumps({
    "kernel": kernel, "pairs_per_launch": pairs, "nn_mode": "grid", "source_points": ns,
    "fetch_size_kb_per_launch": round(fetch), "write_size_kb_per_launch": round(write), "launches_averaged": [nf, nw],
    "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
    "hbm_bytes_per_launch_uncorrected": int((fetch + write) * 1024),
    "algorithmic_bytes_per_launch": int(alg * pairs * ns),
    "compulsory_bytes_of_this_implementation": int(impl * pairs * ns),
    "how": "tools/round_profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and a separate --pmc WRITE_SIZE pass of "
           "tools/profile_target.py B=512 reps=1 (two 256-pair halves on two streams), averaged over the kernel's launches; "
           "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) "
           "prescribes for coalesced multi-dword-per-lane reads on gfx950; WRITE_SIZE is uncalibrated",
}, indent=1))
