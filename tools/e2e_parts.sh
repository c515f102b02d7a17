#!/bin/bash
# the sequence driver with its 256-pair batches as 1, 2 (default), 3 or 4 parts on streams of their own.  usage: tools/e2e_parts.sh
export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_q 1025 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for rep in 1 2; do for p in 1 2 3 4; do
  echo -n "parts $p: "
  $EXE --scans /tmp/drive_q --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --parts $p --out /tmp/pose_p$p.txt 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['pairs_per_s'], d['steady_state_pairs_per_s_rank0'])"
done; done
cmp /tmp/pose_p1.txt /tmp/pose_p2.txt && cmp /tmp/pose_p2.txt /tmp/pose_p3.txt && echo "poses identical"
rm -rf /tmp/drive_q
