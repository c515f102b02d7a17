#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the BENCH workload (512 distinct consecutive pairs of the synthetic drive, extrapolated guesses)
# through tools/fused_probe.py: per-kernel traffic per launch and the traffic of one whole 512-pair step.  usage: r04_step_traffic.sh <tag> [cfg]
tag=${1:-r04x}; cfg=${2:-"fused:"}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
name=${cfg%%:*}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python $R/tools/fused_probe.py pairs=512 distinct=512 steps=2 cfg="$cfg" > $out/pmc_${name}_$c.log 2>&1
done
calib=$out/traffic_calibration.json
[ -f $calib ] || bash $R/tools/traffic_calib.sh $calib > $out/traffic_calibration.log 2>&1
python $R/tools/step_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $calib 5 "tools/fused_probe.py pairs=512 distinct=512 cfg=$cfg: the bench workload (512 distinct consecutive pairs of the synthetic drive, guess = the previous pair's motion), 20 iterations, two 256-pair halves on two streams; 5 identical batches in the run" > $out/step_traffic_$name.json
python $R/tools/pmc_summary.py $out/pmc_FETCH_SIZE nn_ball_lds nn_certify nn_ball_listed accumulate finalize > $out/pmc_${name}_fetch_summary.txt
python $R/tools/pmc_summary.py $out/pmc_WRITE_SIZE nn_ball_lds nn_certify nn_ball_listed accumulate finalize > $out/pmc_${name}_write_summary.txt
if [ $name = fused ]; then
  python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_certify_acc 256 120000 36.8 56 $calib > $out/traffic_nn_certify_acc.json
  python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_lds 256 120000 20 32 $calib > $out/traffic_nn_ball_lds.json
  python $R/tools/traffic_json.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE nn_ball_listed 256 120000 20 20 $calib > $out/traffic_nn_ball_listed.json
fi
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
python -c "import json;d=json.load(open('$out/step_traffic_$name.json'));print('$name step traffic GB', d['step_total_GB'], 'ratio', d['ratio_to_algorithmic']);print({k:(v['launches'],round(v['bytes']/1e9,2)) for k,v in list(d['kernels'].items())[:9]})"
