#!/bin/bash
# quick check of a kernel change: the tests named in $2 (default: the ICP parity files), then fused_probe on the bench batch
# usage: bash tools/r05_quick.sh <tag> "<pytest paths>" "<probe cfg>" [guess kinds]
tag=${1:-r05q}
tests=${2:-tests/test_fused_batch_oracle_gpu.py tests/test_icp_gpu.py}
cfg=${3:-two:;one:no_overlap=1}
kinds=${4:-cv id}
mkdir -p gpurun_out
if [ "$tests" != "none" ]; then
  timeout 900 python -m pytest $tests -m gpu -x -q > gpurun_out/${tag}_pytest.txt 2>&1
  echo "pytest rc=$?" >> gpurun_out/${tag}_pytest.txt
  grep -v amdgpu.ids gpurun_out/${tag}_pytest.txt | tail -15
fi
for g in $kinds; do
  timeout 600 python tools/fused_probe.py pairs=512 distinct=64 steps=6 guess=$g "cfg=$cfg" > gpurun_out/${tag}_probe_$g.txt 2>&1
  grep -v amdgpu.ids gpurun_out/${tag}_probe_$g.txt
done
