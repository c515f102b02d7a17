export TMPDIR=/tmp
python tools/make_drive.py /tmp/drive_q 1025 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
for m in 2 2; do
SMHIP_SHARD_TIMELINE=1 $EXE --scans /tmp/drive_q --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --matchers $m --out /tmp/pose_m.txt 2>&1 | grep -v "^RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | cut -c1-330
done
