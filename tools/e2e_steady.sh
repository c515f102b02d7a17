#!/bin/bash
# the sequence driver over 1 024 and 2 048 pairs of one 2 049-scan drive: whole-run rate, the batch period once the pipeline is full
# (steady_state_pairs_per_s_rank0), seconds, time blocked on the readers
export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_q 2049 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for n in 1024 2048; do
$EXE --scans /tmp/drive_q --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --max-pairs $n --out /tmp/pose.txt 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['pairs'], d['pairs_per_s'], d['steady_state_pairs_per_s_rank0'], d['seconds'], d['wait_for_readers_s_rank0'])"
done
rm -rf /tmp/drive_q
