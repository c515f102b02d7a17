#!/bin/bash
# headline figure against smhip_icp_options.split_after (the iteration from which certify + listed search replace nn_ball_lds;
# 0 = automatic, from the previous batch).  usage: [HEADLINE=identity] split_sweep.sh [N ...]
for s in ${@:-0 1 2 3 5 8}; do
  python bench.py --no-cpu-baseline --no-figures --no-other --headline ${HEADLINE:-extrapolated} --split-after $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('split', $s, 'used', d['config']['split_after'], d['value'], d['kernel_ms_per_step'], d['parity']['worst_trans_err_vs_truth_m'])"
done
