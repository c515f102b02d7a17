#!/bin/bash
# Round-6 closing evidence on the GPU box (one call): the GPU tests, the default bench line, kernel trace + stats of the same command
# with the timed region's per-kernel averages, the calibrated FETCH / WRITE passes over the bench workload (per-kernel traffic records
# dated with the sha256 of the kernel sources, the step's traffic), the single-pair Align (one cooperative launch against the separate
# launches: timings + kernel trace), Ndt config #3 (single + batch of 64: kernel stats, the derivative kernel's traffic record) and
# NdtWithGicp config #5 (kernel stats).  Usage: tools/r06_final.sh <tag> [parts]   (outputs under gpurun_out/<tag>/)
set -u
tag=${1:-r06z}
what=${2:-all}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
want() { [ "$what" = all ] || [[ ",$what," == *",$1,"* ]]; }
if want tests; then
  timeout -k 5 1500 python -m pytest tests -m gpu -q > "$out/pytest_gpu.txt" 2>&1
  tail -3 "$out/pytest_gpu.txt"
fi
if want bench; then
  t0=$SECONDS
  timeout -k 5 900 python bench.py > "$out/bench_line.json" 2> "$out/bench.err" < /dev/null
  echo "default bench.py run: $((SECONDS - t0)) s wall" > "$out/bench_wall.txt"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --no-cpu-baseline --no-end-to-end --no-other > "$out/bench_line_traced.json" 2> "$out/trace.err" < /dev/null
  f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep -v "at::native" "$f" | cut -c1-400 > "$out/bench_kernel_stats.csv"
  # trace averages over the launches of the timed region only (4 untimed steps first, then 5 timed steps; 2 parts of 256 pairs per step)
  P=2
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
  cut -c1-1200 "$out/bench_line.json"; echo; cat "$out/timed_region_trace_average.txt"
fi
if want traffic; then
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
  cd $R
  python tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_certify_acc 256 120000 36.8 52 $calib > $out/traffic_nn_certify_acc.json
  python tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_lds 256 120000 20 36 $calib > $out/traffic_nn_ball_lds.json
  python tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_listed 256 120000 20 20 $calib > $out/traffic_nn_ball_listed.json
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
  python -c "import json;d=json.load(open('$out/step_traffic.json'));print('step traffic GB', d['step_total_GB'], 'ratio', d['ratio_to_algorithmic'])"
fi
if want single; then
  for cfg in "one:" "separate:separate=1"; do
    name=${cfg%%:*}; args=${cfg#*:}
    for g in near far; do
      python tools/single_pair_probe.py guess=$g $args 2>&1 | grep -a "single pair" | sed "s/^/[$name, guess=$g] /" >> "$out/single_pair_probe.txt"
    done
  done
  cat "$out/single_pair_probe.txt"
  bash tools/r06_single_trace.sh $tag/single_one guess=near > /dev/null 2>&1
  bash tools/r06_single_trace.sh $tag/single_separate guess=near separate=1 > /dev/null 2>&1
  grep smhip "$out/single_one/kernel_stats.csv" | head -12 | cut -c1-160
fi
if want ndt; then
  bash tools/r06_ndt_profile.sh $tag/z > "$out/ndt_single.log" 2>&1
  bash tools/r06_ndt_batch_profile.sh $tag/z > "$out/ndt_batch.log" 2>&1
  bash tools/ndt_traffic.sh $tag/z > "$out/ndt_traffic.log" 2>&1
  bash tools/gicp_profile.sh $tag > "$out/gicp.log" 2>&1
  tail -3 "$out/ndt_single.log"; tail -6 "$out/ndt_batch.log" | cut -c1-200; grep -a "ratio_to_algorithmic\|ms_per_launch" "$out/ndt_traffic.log"; tail -6 "$out/gicp.log" | cut -c1-200
fi
