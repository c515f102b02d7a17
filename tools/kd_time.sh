#!/bin/bash
# kd_build / nn_nabo per-launch durations of the reference-search mode (rocprofv3 kernel stats of the timing probe).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kd_prof -o kd -- python $R/tools/gpu_probe.py 120000 64 mode=2 > $R/gpurun_out/kd_probe.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/kd_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5: print(r["Name"][:50], r["Calls"], r["AverageNs"], r["Percentage"])
PY
