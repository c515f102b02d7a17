"""Per-launch durations (us), in launch order, of selected kernels from a rocprofv3 --kernel-trace csv directory.
Usage: trace_iterations.py <dir> <kernel substring> [...]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for want in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if want in r["Kernel_Name"]]
    print(want, "launches", len(d), "total_us", round(sum(d), 1))
    print("  ", " ".join(f"{x:.0f}" for x in d))
