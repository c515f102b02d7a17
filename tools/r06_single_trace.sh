#!/bin/bash
# kernel trace of the single-pair Align (tools/single_pair_probe.py): per-kernel stats and the last Align's per-launch durations
# usage: tools/r06_single_trace.sh <tag> [probe args...]
tag=${1:-r06s}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/tools/single_pair_probe.py "$@" 2>&1 | grep -v amdgpu.ids > $out/probe.txt
rm -rf $out/t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python $R/tools/single_pair_probe.py "$@" > $out/trace.log 2>&1
find $out/t -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
cat $out/probe.txt
grep smhip $out/kernel_stats.csv | head -24 | cut -c1-150
python $R/tools/trace_sequence.py $out/t 28 nn_ball nn_certify listed finalize accumulate iteration_sums nn_refine_one single_ > $out/sequence.txt
cat $out/sequence.txt
rm -rf $out/t
