#!/bin/bash
# PMC passes over the libnabo-mode kernels (64 pairs, one stream): instruction counts per walk / certificate launch, wait
# cycles, LDS conflicts, cache hit rates.  Usage: bash tools/nabo_pmc.sh <outdir-under-gpurun_out> [nocert=1]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-nabo_pmc}
extra=${2:-nocert=0}
mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf $out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python $R/tools/profile_target.py B=64 reps=1 noov=1 mode=2 $extra > $out/p$i.log 2>&1
  python $R/tools/pmc_summary.py $out/p$i nn_nabo nn_certify kd_build 2>&1 | cut -c1-400
  rm -rf $out/p$i
done
