"""Kernel timeline (start offset, duration, gap to the previous kernel's end; us) from a rocprofv3 --kernel-trace csv directory.
Usage: trace_timeline.py <dir> [first_row [n_rows]]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'].split('(')[0][:60]}  grid {r.get('Grid_Size', '?')}")
    prev_end = e
print("rows", len(rows))
