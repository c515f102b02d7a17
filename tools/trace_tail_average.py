"""Average duration of launches [first, first + count) (in launch order) of a kernel in a rocprofv3 --kernel-trace csv
directory.  Usage: trace_tail_average.py <dir> <kernel substring> <first> <count>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sys.argv[2] in r["Kernel_Name"]]
first, count = int(sys.argv[3]), int(sys.argv[4])
part = d[first:first + count]
print(f"{sys.argv[2]}: {len(d)} launches in the trace; launches {first}..{first + len(part) - 1}: average {sum(part) / len(part):.1f} us, "
      f"min {min(part):.1f}, max {max(part):.1f}")
