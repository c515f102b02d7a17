#!/bin/bash
# Round-5 closing evidence on the GPU box: the default bench line, kernel trace + stats of the same command with the timed region's
# per-kernel averages, the calibrated FETCH / WRITE passes over the bench workload (per kernel and summed over one step), the counter
# passes of the first-iterations search kernels.  Usage: tools/r05_final.sh <tag>   (outputs under gpurun_out/<tag>/)
set -u
tag=${1:-r05z}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
t0=$SECONDS
timeout -k 5 480 python bench.py > "$out/bench_line.json" 2> "$out/bench.err" < /dev/null
echo "default bench.py run: $((SECONDS - t0)) s wall" > "$out/bench_wall.txt"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline --no-end-to-end --no-other > "$out/bench_line_traced.json" 2> "$out/trace.err" < /dev/null
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv"
# trace averages over the launches of the timed region only: 4 untimed steps first, then 5 timed steps; a 512-pair step runs as
# P parts (default 2 of 256 pairs), so per step 2 P launches of nn_ball_lds and accumulate (iterations 0-1), 18 P of nn_certify_acc /
# listed_plan / nn_ball_listed_items / nn_refine_one / iteration_sums, 20 P of finalize, P of every grid kernel
P=${PARTS:-2}
: > "$out/timed_region_trace_average.txt"
echo "# rocprofv3 --kernel-trace of bench.py: the launches of the 5 timed steps ($P parts of $((512 / P)) pairs per step)" >> "$out/timed_region_trace_average.txt"
for k in nn_certify_acc nn_ball_listed_items listed_plan nn_refine_one iteration_sums; do
  python tools/trace_tail_average.py "$out/trace" $k $((4 * 18 * P)) $((5 * 18 * P)) >> "$out/timed_region_trace_average.txt" 2>&1
done
for k in nn_ball_lds accumulate; do
  python tools/trace_tail_average.py "$out/trace" $k $((4 * 2 * P)) $((5 * 2 * P)) >> "$out/timed_region_trace_average.txt" 2>&1
done
python tools/trace_tail_average.py "$out/trace" finalize $((4 * 20 * P)) $((5 * 20 * P)) >> "$out/timed_region_trace_average.txt" 2>&1
for k in grid_mark grid_rank grid_count grid_place; do
  python tools/trace_tail_average.py "$out/trace" $k $((4 * P)) $((5 * P)) >> "$out/timed_region_trace_average.txt" 2>&1
done
rm -rf "$out/trace"
# FETCH_SIZE / WRITE_SIZE passes over the bench workload (512 distinct pairs) through tools/fused_probe.py
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python $R/tools/fused_probe.py pairs=512 distinct=512 steps=2 cfg="fused:" > $out/pmc_$c.log 2>&1
done
calib=$out/traffic_calibration.json
[ -f $calib ] || bash $R/tools/traffic_calib.sh $calib > $out/traffic_calibration.log 2>&1
python $R/tools/step_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $calib 5 "tools/fused_probe.py pairs=512 distinct=512 cfg=fused: the bench workload (512 distinct consecutive pairs of the synthetic drive, guess = the previous pair's motion), 20 iterations, two 256-pair halves on two streams; 5 identical batches in the run" > $out/step_traffic.json
python $R/tools/pmc_summary.py $out/pmc_FETCH_SIZE nn_ball_lds nn_certify nn_ball_listed accumulate finalize iteration_sums grid_ > $out/pmc_fetch_summary.txt
python $R/tools/pmc_summary.py $out/pmc_WRITE_SIZE nn_ball_lds nn_certify nn_ball_listed accumulate finalize iteration_sums grid_ > $out/pmc_write_summary.txt
# (implementation bytes per point of the fused certificate pass: 12 source + 4 shadow read, 4 distance written, 32 gathered x the kept share)
python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_certify_acc 256 120000 36.8 52 $calib > $out/traffic_nn_certify_acc.json
python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_lds 256 120000 20 36 $calib > $out/traffic_nn_ball_lds.json
python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_listed 256 120000 20 20 $calib > $out/traffic_nn_ball_listed.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
cd $R
bash tools/r05_search_pmc.sh $tag/search_pmc > /dev/null 2>&1
cut -c1-1500 "$out/bench_line.json"; echo; cat "$out/timed_region_trace_average.txt"
python -c "import json;d=json.load(open('$out/step_traffic.json'));print('step traffic GB', d['step_total_GB'], 'ratio', d['ratio_to_algorithmic']);print({k:(v['launches'],round(v['bytes']/1e9,2)) for k,v in list(d['kernels'].items())[:10]})"
