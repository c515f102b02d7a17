#!/bin/bash
# the sequence driver with one matcher against two taking the batches in turn, same box, same files.  usage: tools/e2e_ab.sh [n_scans]
N=${1:-1025}
export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_ab $N 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for rep in 1 2 3; do
  for m in 1 2; do
    echo -n "matchers=$m: "
    $EXE --scans /tmp/drive_ab --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --matchers $m --out /tmp/pose_$m.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['pairs_per_s'], 'prep', d['prepare_targets_s_rank0'], 'upload', d['upload_s_rank0'], 'total', d['seconds'])"
  done
done
cmp /tmp/pose_1.txt /tmp/pose_2.txt && echo "poses identical"
rm -rf /tmp/drive_ab
