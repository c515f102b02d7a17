#!/bin/bash
# the one-launch Align under stress at HEAD (one file for profiles/): repeated Aligns of small clouds on grids of mostly idle workgroups,
# single pairs and mixes of 2-8 pairs of different sizes, with and without the early exit, every Align compared with the first bit for
# bit; six matchers on six threads.  usage: tools/r06_one_launch_stress.sh <out file>
out=${1:-gpurun_out/one_launch_stress.txt}
mkdir -p $(dirname $out); : > $out
run() { echo "\$ $*" >> $out; timeout 600 "$@" 2>&1 | tail -3 >> $out; }
for n in 257 1000 5000; do run python tools/one_race_probe.py n=$n reps=200 pairs=1 early=1; done
run python tools/one_race_probe.py n=5000 reps=200 pairs=1 early=0
for cfg in "pairs=2 mix=ab" "pairs=3 mix=aab" "pairs=4 mix=ab" "pairs=5 mix=abb" "pairs=6 mix=ab" "pairs=7 mix=abb" "pairs=8 mix=aab" "pairs=8 mix=ab"; do
  for early in 1 0; do run python tools/one_race_probe.py n=5000 reps=200 $cfg early=$early; done
done
run python tools/one_stress.py threads=6 reps=20
cat $out
