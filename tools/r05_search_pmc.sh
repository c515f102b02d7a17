#!/bin/bash
# Counter passes over the every-query-searches kernels (64 replicated pairs, one stream): nn_ball_lds (SMHIP_WAVE_SEARCH=0) against
# nn_ball_wave (=1): instruction mix, busy / wait cycles, LDS conflicts.  usage: tools/r05_search_pmc.sh <tag>
tag=${1:-r05pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
: > $out/summary.txt
for ws in 0 1; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rm -rf $out/p
    SMHIP_WAVE_SEARCH=$ws timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p -- python $R/tools/profile_target.py B=64 reps=1 noov=1 > $out/p_${ws}_$i.log 2>&1
    echo "== SMHIP_WAVE_SEARCH=$ws set $i" >> $out/summary.txt
    python $R/tools/pmc_summary.py $out/p nn_ball_lds nn_ball_wave 2>&1 | cut -c1-600 >> $out/summary.txt
    rm -rf $out/p
  done
done
cat $out/summary.txt
