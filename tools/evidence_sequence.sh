# BASELINE config #4 end to end with the C++ driver: 4541-pose synthetic drive -> KITTI .bin files -> smhip_shard -> kitti_pose.txt
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out/${TAG}_seq
df -h /tmp | tail -1
N=${N:-4541}; PTS=${PTS:-120000}
avail_kb=$(df --output=avail /tmp | tail -1)
need_kb=$(( N * PTS * 16 / 1024 + 1048576 ))
if [ "$avail_kb" -lt "$need_kb" ]; then PTS=40000; echo "not enough room in /tmp for 120k-point scans: using $PTS points per scan"; fi
t0=$(date +%s)
timeout 1500 python tools/make_drive.py /tmp/drive $N $PTS --cuda 2>&1 | tail -2
t1=$(date +%s); echo "generation wall: $((t1 - t0)) s"
EXE=staticmapping_amd/lib/smhip_shard
for mode in "fixed20 --iterations 20 --early-exit 0" "earlyexit --iterations 100 --early-exit 1"; do
  set -- $mode; name=$1; shift
  timeout 900 $EXE --scans /tmp/drive --gpus 1 --batch 64 --guess-tx 0.8 --out gpurun_out/${TAG}_seq/kitti_pose_$name.txt "$@" > gpurun_out/${TAG}_seq/driver_$name.json 2> gpurun_out/${TAG}_seq/driver_$name.err
  echo "driver $name rc=$?"; cat gpurun_out/${TAG}_seq/driver_$name.json
  python tools/sequence_eval.py gpurun_out/${TAG}_seq/kitti_pose_$name.txt /tmp/drive_truth.txt | tee gpurun_out/${TAG}_seq/eval_$name.json
done
# identity guess, as SURVEY cfg 4 words it (many pairs fall into the wrong basin: the oracle does too)
timeout 900 $EXE --scans /tmp/drive --gpus 1 --batch 64 --guess-tx 0.0 --iterations 20 --out gpurun_out/${TAG}_seq/kitti_pose_identity.txt > gpurun_out/${TAG}_seq/driver_identity.json 2>&1
cat gpurun_out/${TAG}_seq/driver_identity.json
python tools/sequence_eval.py gpurun_out/${TAG}_seq/kitti_pose_identity.txt /tmp/drive_truth.txt | tee gpurun_out/${TAG}_seq/eval_identity.json
echo "points per scan: $PTS" > gpurun_out/${TAG}_seq/workload.txt
rm -rf /tmp/drive
