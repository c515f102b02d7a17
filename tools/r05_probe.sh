#!/bin/bash
# round 5 working probe: the GPU tests, then the bench batch's per-class composition (two streams / one stream, extrapolated and
# identity guesses) through tools/fused_probe.py.  usage: bash tools/r05_probe.sh <tag> [notests]
tag=${1:-r05a}
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu.txt
  tail -5 gpurun_out/${tag}_pytest_gpu.txt
fi
timeout 600 python tools/fused_probe.py pairs=512 distinct=64 steps=6 guess=cv "cfg=two:;one:no_overlap=1;separate:no_fused_sums=1" > gpurun_out/${tag}_probe_cv.txt 2>&1
cat gpurun_out/${tag}_probe_cv.txt | grep -v amdgpu.ids
timeout 600 python tools/fused_probe.py pairs=512 distinct=64 steps=4 guess=id "cfg=two:;one:no_overlap=1" > gpurun_out/${tag}_probe_id.txt 2>&1
cat gpurun_out/${tag}_probe_id.txt | grep -v amdgpu.ids
