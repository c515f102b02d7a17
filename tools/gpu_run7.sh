set -u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ndt_gicp_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -5
python tools/gicp_probe.py 2>&1 | tail -3
