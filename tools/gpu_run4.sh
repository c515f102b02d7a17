set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/ndt_probe.py 2>&1 | tail -3
python tools/single_pair_probe.py 2>&1 | tail -4
python tools/gicp_probe.py 2>&1 | tail -3
