#!/bin/bash
# Round-4 evidence run on the GPU box: the bench line, kernel trace + stats of the same command, calibrated FETCH / WRITE passes
# (per kernel and summed over one step), per-iteration durations on one stream.  Usage: tools/r04_profile.sh <tag>
set -u
tag=${1:-r04x}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
t0=$SECONDS
python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
echo "default bench.py run: $((SECONDS - t0)) s wall" > "$out/bench_wall.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline --no-end-to-end --no-other > "$out/bench_line_traced.json" 2> "$out/trace.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/bench_kernel_stats.csv" \;
# trace averages over the launches of the timed region only: 4 untimed steps first (one plain and one fully profiled step that pick
# the kernel class to bracket, 2 warm-up steps), then 5 timed steps; per step 4 launches of nn_ball_lds (iterations 0-1 x 2 halves)
# and 36 of nn_certify_acc / nn_ball_listed, 40 of accumulate (36 of them the early-exit form) and finalize
python tools/trace_tail_average.py "$out/trace" nn_certify_acc 144 180 > "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" nn_ball_lds 16 20 >> "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" nn_ball_listed 144 180 >> "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" finalize 160 200 >> "$out/timed_region_trace_average.txt"
python tools/trace_tail_average.py "$out/trace" accumulate 160 200 >> "$out/timed_region_trace_average.txt"
rm -rf "$out/trace"
bash tools/traffic_calib.sh "$out/traffic_calibration.json" > "$out/traffic_calibration.log" 2>&1
calib="$out/traffic_calibration.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_write.log" 2>&1
python tools/pmc_summary.py "$out/pmc_fetch" nn_ball_lds nn_certify nn_ball_listed accumulate finalize > "$out/pmc_fetch_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" nn_ball_lds nn_certify nn_ball_listed accumulate finalize > "$out/pmc_write_summary.txt"
# per source point (algorithmic = SURVEY 8(d); implementation = what this code must move):
#   nn_certify_acc  reads 12 B point + 4 B previous match + 4 B bound, writes 4 B d2, gathers the 16-byte match and its 16-byte normal: 56 B;
#                   FindClosests 20 B + ErrorElements / ComputePointToPlane 24 rho = 36.8 B
#   nn_ball_lds     reads the same 20 B and writes id, d2, bound: 32 B; FindClosests = 20 B
#   nn_ball_listed  a few thousand scattered queries per pair: priced with the full 20 B/pt of FindClosests, which flatters it (see calib_scatter)
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_certify_acc 256 120000 36.8 56 "$calib" > "$out/traffic_nn_certify_acc.json"
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_ball_lds 256 120000 20 32 "$calib" > "$out/traffic_nn_ball_lds.json"
python tools/traffic_json.py "$out/pmc_fetch" "$out/pmc_write" nn_ball_listed 256 120000 20 20 "$calib" > "$out/traffic_nn_ball_listed.json"
python tools/step_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$calib" > "$out/step_traffic.json"
rm -rf "$out/pmc_fetch" "$out/pmc_write"
# the listed search's cache behaviour: L2 requests / hits / misses and L1 -> L2 read requests
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$out/pmc_tcc" -- python tools/profile_target.py B=512 reps=1 > "$out/pmc_tcc.log" 2>&1
python tools/pmc_summary.py "$out/pmc_tcc" nn_ball_listed nn_certify_acc nn_ball_lds finalize > "$out/pmc_tcc_summary.txt"
rm -rf "$out/pmc_tcc"
cat "$out/bench_line.json" | cut -c1-1500; echo; cat "$out/timed_region_trace_average.txt"; cat "$out/pmc_fetch_summary.txt" "$out/pmc_write_summary.txt" "$out/pmc_tcc_summary.txt"; head -12 "$out/bench_kernel_stats.csv" | cut -c1-160
python -c "import json;d=json.load(open('$out/step_traffic.json'));print('step traffic GB', d['step_total_GB'], 'ratio', d['ratio_to_algorithmic']);print({k:(v['launches'],round(v['bytes']/1e9,2)) for k,v in list(d['kernels'].items())[:8]})"
# per-iteration durations of one 64-pair batch on ONE stream (no overlap) for reading the iteration profile
rocprofv3 --kernel-trace --output-format csv -d "$out/seq" -- python tools/profile_target.py B=64 reps=1 noov=1 > "$out/seq.log" 2>&1
python tools/trace_iterations.py "$out/seq" nn_ball_lds nn_certify_acc nn_ball_listed accumulate finalize nn_refine_one nn_validate > "$out/iteration_profile.txt"
rm -rf "$out/seq"
head -30 "$out/iteration_profile.txt"
