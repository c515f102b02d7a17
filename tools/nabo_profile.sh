#!/bin/bash
# per-iteration kernel durations of one batched IcpFast step in the reference's search mode (nn_mode NABO), one stream
tag=${1:-x}
out=gpurun_out/nabo_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python tools/fused_probe.py pairs=512 distinct=512 steps=1 cfg="nabo_one:nn_mode=2,no_overlap=1" > $out/nabo_trace.log 2>&1 < /dev/null
python tools/trace_sequence.py $out/t 20 "nn_nabo<1, true" "nn_certify<20, true" "accumulate<" "nn_nabo<4, false" finalize kd_build > $out/nabo_sequence.txt 2>&1
rm -rf $out/t
cat $out/nabo_sequence.txt
