set -u
export TMPDIR=/tmp
bash tools/round_profile.sh r02b > gpurun_out/r02b_profile.log 2>&1
tail -40 gpurun_out/r02b_profile.log
for sp in 4 5 6 8; do python tools/gpu_probe.py 120000 64,512 split=$sp 2>&1 | grep -v "^   \|^ns"; done
