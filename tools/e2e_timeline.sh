#!/bin/bash
# where the host is as the sequence driver walks its batches (SMHIP_SHARD_TIMELINE), with and without the warm-up batch before the clock
export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_q ${1:-1025} 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for w in 1 1 0; do
  SMHIP_SHARD_TIMELINE=1 $EXE --scans /tmp/drive_q --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --warmup $w --out /tmp/pose_w$w.txt 2>&1 | grep -v "^RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | cut -c1-420
done
cmp /tmp/pose_w0.txt /tmp/pose_w1.txt && echo "poses identical"
rm -rf /tmp/drive_q
