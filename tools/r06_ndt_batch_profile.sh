# NDT config #3 batch of 64: kernel stats of the lock-step batch (rebuilt + kept).  usage: r06_ndt_batch_profile.sh <tag>
set -u
export TMPDIR=/tmp
tag=${1:-r06a}
out=$PWD/gpurun_out/${tag}_ndt_batch
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/ndt_cfg3_probe.py K=64 reps=2 distinct=8 > "$out/probe_traced.txt" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
python tools/trace_timeline.py "$out/trace" 0 100000 > "$out/timeline_all.txt"
grep -v "at::native\|elementwise\|reduce_kernel" "$out/timeline_all.txt" | tail -260 > "$out/timeline_tail.txt"
rm -rf "$out/trace" "$out/timeline_all.txt"
cat "$out/probe_traced.txt" | tail -4
grep -v "at::native" "$out/kernel_stats.csv" | head -30 | cut -c1-150
