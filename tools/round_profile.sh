#!/bin/bash
# Round-end evidence run on the GPU box: kernel trace + stats of bench.py, the plain bench line, and the
# separate FETCH_SIZE / WRITE_SIZE passes for the dominant kernel.  Usage: tools/round_profile.sh <tag>
set -u
tag=${1:-r01x}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline > "$out/bench_line_traced.json" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/bench_kernel_stats.csv" \;
# trace average of the dominant kernel over the launches of the timed region only: 16 launches per step (8 iterations x 2
# half-batches), 2 warm-up steps first, 5 timed steps, then the untimed breakdown / alone steps that the --stats average
# above also covers.  This is the number roofline.avg_launch_ms of the traced line must match.
python tools/trace_tail_average.py "$out/trace" nn_ball_lds 32 80 > "$out/timed_region_trace_average.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_write.log" 2>&1
python tools/pmc_summary.py "$out/pmc_fetch" nn_ball_lds nn_certify accumulate > "$out/pmc_fetch_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" nn_ball_lds nn_certify accumulate > "$out/pmc_write_summary.txt"
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
cat "$out/bench_line.json"; cat "$out/timed_region_trace_average.txt"; cat "$out/pmc_fetch_summary.txt" "$out/pmc_write_summary.txt"; head -8 "$out/bench_kernel_stats.csv"
# per-iteration durations of one 64-pair batch on ONE stream (no overlap) for reading the iteration profile
rocprofv3 --kernel-trace --output-format csv -d "$out/seq" -- python tools/profile_target.py B=64 reps=1 noov=1 > "$out/seq.log" 2>&1
python tools/trace_iterations.py "$out/seq" nn_ball_lds nn_certify nn_ball_listed accumulate finalize nn_validate "nn_ring" nn_fallback > "$out/iteration_profile.txt"
rm -rf "$out/seq"
cat "$out/iteration_profile.txt"
