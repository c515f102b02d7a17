#!/bin/bash
# kernel statistics of NdtWithGicp at config #5: single kept, batch of 16 rebuilt, batch of 16 kept (GPU box; writes gpurun_out/gicp_prof_<tag>/)
tag=${1:-x}
out=gpurun_out/gicp_prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "single_kept:jobs=1 single=1 mode=kept reps=1" "batch_rebuilt:single=0 mode=rebuilt reps=3" "batch_kept:single=0 mode=kept reps=3"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o p -- python tools/gicp_batch_probe.py cells=0.5 $args > $out/$name.log 2>&1 < /dev/null
  f=$(ls $out/$name/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cut -d, -f1-4 "$f" | cut -c1-150 | head -22 > $out/$name.top.txt; cp "$f" $out/$name.kernel_stats.csv; fi
  rm -rf $out/$name
  grep -v amdgpu.ids $out/$name.log | tail -4
done
