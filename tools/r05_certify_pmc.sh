#!/bin/bash
# Counter passes over the kernels of the fused (settled) iteration, 64 replicated pairs on one stream (tools/profile_target.py):
# instruction mix, busy / wait cycles, L2 hits and misses of nn_certify_acc, nn_ball_listed_items, iteration_sums and finalize,
# averaged over the launches of one alignment batch.  usage: tools/r05_certify_pmc.sh <tag>
tag=${1:-r05cpmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
: > $out/summary.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $out/p
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p -- python $R/tools/profile_target.py B=64 reps=1 noov=1 > $out/p_$i.log 2>&1
  echo "== set $i" >> $out/summary.txt
  python $R/tools/pmc_summary.py $out/p nn_certify_acc nn_ball_listed_items iteration_sums finalize 2>&1 | cut -c1-600 >> $out/summary.txt
  rm -rf $out/p
done
cat $out/summary.txt
