"""Kernels and memory copies of a rocprofv3 trace directory between two offsets (ms from the first kernel's start), one line each,
consecutive launches of the same kernel folded.  Usage: trace_window.py <dir> <from_ms> <to_ms>"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("smhip::", "").replace("(anonymous namespace)::", "")[:44] + " q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
t0 = min(s for s, _, n in ev if n.startswith("K"))
a, b = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
last = None; cnt = 0; first_s = 0; last_e = 0; tot = 0
def flush():
    if last is not None: print(f"{(first_s - t0) / 1e6:9.3f} .. {(last_e - t0) / 1e6:9.3f} ms  x{cnt:<4d} sum {tot / 1e6:7.3f} ms  {last}")
for s, e, n in ev:
    if s < a or s > b: continue
    key = n if n.startswith("K") else n.split(" ")[0] + " " + n.split(" ")[1]
    if key != last:
        flush(); last = key; cnt = 0; first_s = s; tot = 0
    cnt += 1; last_e = e; tot += e - s
flush()
