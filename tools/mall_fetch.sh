#!/bin/bash
# HBM fetch per pair of the streaming iteration kernels at batch sizes that do / do not fit the 256 MiB Infinity Cache
# (VERDICT r2 3(d)): FETCH_SIZE per launch / pairs per launch, calibrated factor x 2.000 (profiles/r03p_traffic_calibration.json)
set -u
export TMPDIR=/tmp
R=$PWD
cd /tmp
for B in 32 64 128 512; do
  rm -rf /tmp/mf_$B
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/mf_$B -- python $R/tools/profile_target.py B=$B reps=1 > /tmp/mf_$B.log 2>&1
  echo "== batch of $B pairs (launches of $((B / 2)) pairs on two streams; FETCH_SIZE in KB per launch)"
  python $R/tools/pmc_summary.py /tmp/mf_$B "accumulate<" nn_certify | python -c "
import sys, re
for ln in sys.stdin:
    m = re.search(r\"'FETCH_SIZE': (\d+)\", ln)
    name = ln.split(' {')[0][:40]
    if m: print('   %-40s %10d KB per launch = %8.1f KB per pair (x 2.000 = %.2f MB)' % (name, int(m.group(1)), int(m.group(1)) / max(1, $B // 2), 2 * int(m.group(1)) / max(1, $B // 2) / 1e3))
"
done
