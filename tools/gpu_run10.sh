set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r02_gicp2
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/gicp_probe.py > "$out/probe.txt" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/gicp_kernel_stats.csv" \;
rm -rf "$out/trace"
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_gicp2/gicp_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:80]:82s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
