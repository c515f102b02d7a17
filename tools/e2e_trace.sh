#!/bin/bash
# kernel trace + stats of the sequence driver's end-to-end run (files -> poses).  usage: tools/e2e_trace.sh <tag> [n_scans]
tag=${1:-e2e}; N=${2:-1025}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/make_drive.py /tmp/drive_tr $N 120000 --cuda 2>&1 | tail -1
EXE=staticmapping_amd/lib/smhip_shard
$EXE --scans /tmp/drive_tr --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --out /tmp/pose_tr.txt | tail -1 > $out/run_untraced.json
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -- $EXE --scans /tmp/drive_tr --gpus 1 --guess-tx 0.8 --iterations 20 --early-exit 0 --out /tmp/pose_tr.txt 2> $out/trace.err | tail -1 > $out/run_traced.json
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
python - $f > $out/kernel_stats.txt <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:36]:
    print(r[0].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60].ljust(60), r[1].rjust(5), '%8.2f ms' % (float(r[2]) / 1e6), '%9.1f us' % (float(r[3]) / 1e3))
PY
python tools/trace_busy.py $out/trace 10 > $out/busy.txt
python tools/trace_window.py $out/trace ${3:-360} ${4:-400} > $out/window.txt
rm -rf $out/trace /tmp/drive_tr
cut -c1-330 $out/run_untraced.json; head -12 $out/kernel_stats.txt; tail -22 $out/busy.txt; cat $out/window.txt
