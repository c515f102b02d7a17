"""Durations (us) of the last `count` launches of every kernel whose name contains one of the given substrings, in launch order,
from a rocprofv3 --kernel-trace csv directory.  Usage: trace_sequence.py <dir> <count> <substring> [<substring> ...]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2])
for sub in sys.argv[3:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sub in r["Kernel_Name"]]
    print(f"{sub}: {len(d)} launches; last {min(count, len(d))}: " + " ".join(f"{x:.0f}" for x in d[-count:]))
