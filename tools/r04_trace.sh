#!/bin/bash
# kernel-trace statistics of the batched bench workload through tools/fused_probe.py: two streams (default) and one stream
# (every kernel alone on the GPU).  usage: tools/r04_trace.sh <tag> [cfg for fused_probe]
tag=${1:-r04x}
cfg=${2:-"fused:"}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in two one; do
  c="$cfg"; [ $mode = one ] && c="${cfg%:*}:no_overlap=1"
  rm -rf $out/t_$mode
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$mode -- python $R/tools/fused_probe.py pairs=512 steps=2 cfg="$c" > $out/trace_$mode.log 2>&1
  find $out/t_$mode -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_$mode.csv \;
  rm -rf $out/t_$mode
  tail -3 $out/trace_$mode.log
  head -22 $out/kernel_stats_$mode.csv | cut -c1-160
done
