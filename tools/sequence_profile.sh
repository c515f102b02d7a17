#!/bin/bash
# kernel stats of the C++ sequence driver over a short synthetic drive: where the per-scan time in front of the alignment goes
set -u
export TMPDIR=/tmp
R=$PWD
N=${N:-513}
python tools/make_drive.py /tmp/drive_p $N 120000 --cuda 2>&1 | tail -1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/seq_prof -o seq -- $R/staticmapping_amd/lib/smhip_shard --scans /tmp/drive_p --gpus 1 --guess-tx 0.8 --iterations 20 --out /tmp/kp.txt > $R/gpurun_out/seq_prof.json 2> $R/gpurun_out/seq_prof.err
cd $R
tail -1 gpurun_out/seq_prof.json | cut -c1-300
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/seq_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f ms" % (tot / 1e6))
    for r in rows[:22]:
        print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), "%8.1f ms" % (float(r["TotalDurationNs"]) / 1e6), "%8.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
rm -rf /tmp/drive_p gpurun_out/seq_prof
