#!/bin/bash
# kernel-trace stats of the reference-search batch (tools/nabo_probe.py): per-kernel totals of three 512-pair batches + one profiled
set -u
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/nabo_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nabo_kt -o nk -- python $R/tools/nabo_probe.py > /tmp/nabo_kt.log 2>&1
grep -v amdgpu /tmp/nabo_kt.log | cut -c1-220 | head -3
cd $R
python - <<PY
import csv, glob
for f in glob.glob("/tmp/nabo_kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f ms" % (tot / 1e6))
    for r in rows[:14]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%8.1f ms" % (float(r["TotalDurationNs"]) / 1e6), "%8.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
