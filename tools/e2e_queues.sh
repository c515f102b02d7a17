#!/bin/bash
# the sequence driver with the runtime's default number of hardware queues (4) against 8 and 16: does the upload stream of the next batch
# get a queue of its own?  usage: tools/e2e_queues.sh [n_scans]
N=${1:-1025}
export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_q $N 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for rep in 1 2 3; do
  for q in default 8 16; do
    echo -n "GPU_MAX_HW_QUEUES=$q: "
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    $EXE --scans /tmp/drive_q --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 ${E2E_EXTRA:-} --out /tmp/pose_$q.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['pairs_per_s'], 'prep', d['prepare_targets_s_rank0'], 'upload', d['upload_s_rank0'], 'total', d['seconds'])"
  done
done
cmp /tmp/pose_default.txt /tmp/pose_8.txt && echo "poses identical"
rm -rf /tmp/drive_q
