#!/bin/bash
# one-stream kernel trace of the bench batch through tools/fused_probe.py: per-kernel stats + the last step's per-launch durations
# usage: tools/r05_trace.sh <tag> [cfg] [guess] [kernel substrings...]
tag=${1:-r05t}
cfg=${2:-"one:no_overlap=1"}
guess=${3:-cv}
subs=${4:-iteration_sums finalize accumulate nn_refine_one listed_plan nn_ball_listed_items nn_certify_acc nn_ball}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/t
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python $R/tools/fused_probe.py pairs=512 distinct=32 steps=2 guess=$guess cfg="$cfg" > $out/trace.log 2>&1
find $out/t -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
grep -v amdgpu.ids $out/trace.log | tail -3
grep smhip $out/kernel_stats.csv | head -24 | cut -c1-150
python $R/tools/trace_sequence.py $out/t 40 $subs > $out/sequence.txt
cat $out/sequence.txt
rm -rf $out/t
