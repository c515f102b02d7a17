#!/bin/bash
# timing experiments on nn_certify (SMHIP_DEBUG_FLAGS: 1 = no histogram atomics, 2 = no gather of the previous match, 4 = f32 transforms)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in ${@:-0 1 2 4 7}; do
  rm -rf $R/gpurun_out/cx
  SMHIP_DEBUG_FLAGS=$f rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cx -- python $R/tools/profile_target.py B=64 reps=1 noov=1 > /dev/null 2>&1
  echo "flags $f"; python $R/tools/trace_iterations.py $R/gpurun_out/cx nn_certify accumulate | cut -c1-150
done
rm -rf $R/gpurun_out/cx
