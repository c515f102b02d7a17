#!/bin/bash
# headline figure against the first-iteration search radius.  usage: [HEADLINE=identity] radius_sweep.sh [R ...]
for r in ${@:-0.3 0.2 0.15 0.1}; do
  python bench.py --no-cpu-baseline --no-figures --no-other --headline ${HEADLINE:-extrapolated} --ball-radius $r 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('radius', $r, d['value'], d['kernel_ms_per_step'], d['parity']['worst_trans_err_vs_truth_m'])"
done
