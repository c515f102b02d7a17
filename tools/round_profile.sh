#!/bin/bash
# Round-end evidence run on the GPU box: kernel trace + stats of bench.py, the plain bench line, and the
# separate FETCH_SIZE / WRITE_SIZE passes for the dominant kernel.  Usage: tools/round_profile.sh <tag>
set -u
tag=${1:-r01x}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
t0=$SECONDS
python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
echo "default bench.py run: $((SECONDS - t0)) s wall" > "$out/bench_wall.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline --no-end-to-end > "$out/bench_line_traced.json" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/bench_kernel_stats.csv" \;
# trace average of the iteration kernels over the launches of the timed region only (4 untimed steps first -- one plain and one
# fully profiled step that pick the kernel class to bracket, 2 warm-up steps -- then 5 timed steps, then the untimed breakdown /
# alone steps that the --stats average above also covers).  This is the number
# roofline.avg_launch_ms (and roofline.other_kernels) of the traced line must match.
# (split_after 2: per step 4 launches of nn_ball_lds -- iterations 0-1 x 2 half-batches -- and 36 of nn_certify)
python tools/trace_tail_average.py "$out/trace" nn_certify 144 180 > "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" nn_ball_lds 16 20 >> "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" accumulate 160 200 >> "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" nn_ball_listed 144 180 >> "$out/timed_region_trace_average.txt"
bash tools/traffic_calib.sh "$out/traffic_calibration.json" > "$out/traffic_calibration.log" 2>&1
calib="$out/traffic_calibration.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_write.log" 2>&1
python tools/pmc_summary.py "$out/pmc_fetch" nn_ball_lds nn_certify nn_ball_listed accumulate > "$out/pmc_fetch_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" nn_ball_lds nn_certify nn_ball_listed accumulate > "$out/pmc_write_summary.txt"
# per source point (algorithmic = SURVEY 8(d); implementation = what this code must move):
#   nn_certify   reads 12 B point + 4 B previous match + 4 B bound, writes 4 B d2, gathers the 16-byte match: 40 B; FindClosests = 20 B
#   nn_ball_lds  reads the same 20 B and writes id, d2, bound: 32 B; FindClosests = 20 B
#   accumulate   re-reads 12 B point + 4 B id + 4 B d2 and gathers 32 B (point + normal) for the kept 70 %: 42.4 B; ErrorElements = 24 rho = 16.8 B
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_certify 256 120000 20 40 "$calib" > "$out/traffic_nn_certify.json"
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_ball_lds 256 120000 20 32 "$calib" > "$out/traffic_nn_ball_lds.json"
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_ball_listed 256 120000 20 20 "$calib" > "$out/traffic_nn_ball_listed.json"
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" accumulate 256 120000 16.8 42.4 "$calib" > "$out/traffic_accumulate.json"
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
cat "$out/bench_line.json"; cat "$out/timed_region_trace_average.txt"; cat "$out/pmc_fetch_summary.txt" "$out/pmc_write_summary.txt"; head -8 "$out/bench_kernel_stats.csv"
# per-iteration durations of one 64-pair batch on ONE stream (no overlap) for reading the iteration profile
rocprofv3 --kernel-trace --output-format csv -d "$out/seq" -- python tools/profile_target.py B=64 reps=1 noov=1 > "$out/seq.log" 2>&1
python tools/trace_iterations.py "$out/seq" nn_ball_lds nn_certify nn_ball_listed accumulate finalize nn_validate "nn_ring" nn_fallback > "$out/iteration_profile.txt"
rm -rf "$out/seq"
cat "$out/iteration_profile.txt"
