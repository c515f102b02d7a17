"""GPU occupancy of a rocprofv3 --kernel-trace run over time: per window of the run, the fraction of wall time with at least one kernel
running, the mean number of kernels running, and the kernel classes that hold the most time in it.
Usage: trace_busy.py <dir> [window_ms=10] [prep_name_substrings=kd_,morton,finite_min,rocprim]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("smhip::", "").replace("(anonymous namespace)::", "")[:40]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
print(f"kernels {len(iv)}, first start to last end {(t1 - t0) / 1e6:.1f} ms, sum of durations {sum(e - s for s, e, _ in iv) / 1e6:.1f} ms")
ev = sorted([(s, 1) for s, _, _ in iv] + [(e, -1) for _, e, _ in iv])
busy = collections.defaultdict(float); depth_t = collections.defaultdict(float)
d = 0; prev = t0
for t, k in ev:
    if t > prev:
        a = prev
        while a < t:                                   # split over windows
            w = int((a - t0) // win); b = min(t, t0 + (w + 1) * win)
            if d > 0: busy[w] += b - a
            depth_t[w] += d * (b - a)
            a = b
    d += k; prev = t
cls = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, n in iv:
    cls[int((s - t0) // win)][n] += e - s
tb = 0.0
for w in range(int((t1 - t0) // win) + 1):
    span = min(win, t1 - t0 - w * win)
    top = sorted(cls[w].items(), key=lambda x: -x[1])[:4]
    tb += busy[w]
    print(f"{w * win / 1e6:7.1f} ms: busy {100 * busy[w] / span:5.1f} %  mean kernels running {depth_t[w] / span:4.2f}   " + ", ".join(f"{n} {v / 1e6:.1f}" for n, v in top))
print(f"busy overall {100 * tb / (t1 - t0):.1f} % of {(t1 - t0) / 1e6:.1f} ms")
