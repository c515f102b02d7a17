# NDT config #3: timings, kernel stats and the kernel timeline of the last single Aligns.  usage: r06_ndt_profile.sh <tag> [K]
set -u
export TMPDIR=/tmp
tag=${1:-r06a}; K=${2:-0}
out=$PWD/gpurun_out/${tag}_ndt
mkdir -p "$out"
python tools/ndt_cfg3_probe.py K=$K > "$out/probe.txt" 2>&1
cat "$out/probe.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/ndt_cfg3_probe.py K=0 reps=3 > "$out/probe_traced.txt" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
python tools/trace_timeline.py "$out/trace" 0 100000 > "$out/timeline_all.txt"
tail -150 "$out/timeline_all.txt" > "$out/timeline_tail.txt"
rm -rf "$out/trace" "$out/timeline_all.txt"
head -25 "$out/kernel_stats.csv" | cut -c1-140
