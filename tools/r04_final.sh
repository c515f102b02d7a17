#!/bin/bash
# Round-4 closing evidence on the GPU box: the default bench line, kernel trace + stats of the same command, the calibrated
# FETCH / WRITE passes over the bench workload (per kernel and summed over one step).  Usage: tools/r04_final.sh <tag>
set -u
tag=${1:-r04z}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
t0=$SECONDS
timeout -k 5 480 python bench.py > "$out/bench_line.json" 2> "$out/bench.err" < /dev/null
echo "default bench.py run: $((SECONDS - t0)) s wall" > "$out/bench_wall.txt"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline --no-end-to-end --no-other > "$out/bench_line_traced.json" 2> "$out/trace.err" < /dev/null
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv"
# trace averages over the launches of the timed region only: 4 untimed steps first, then 5 timed steps; per step 4 launches of
# nn_ball_lds (iterations 0-1 x 2 halves), 36 of nn_certify_acc / listed_plan / nn_ball_listed_items, 40 of accumulate and finalize
for spec in "nn_certify_acc 144 180" "nn_ball_lds 16 20" "nn_ball_listed_items 144 180" "listed_plan 144 180" "finalize 160 200" "accumulate 160 200"; do
  python tools/trace_tail_average.py "$out/trace" $spec >> "$out/timed_region_trace_average.txt" 2>&1
done
rm -rf "$out/trace"
timeout -k 5 700 bash tools/r04_step_traffic.sh $tag "fused:" > "$out/step_traffic.log" 2>&1 < /dev/null
cut -c1-1200 "$out/bench_line.json"; echo; cat "$out/timed_region_trace_average.txt"; tail -3 "$out/step_traffic.log"
