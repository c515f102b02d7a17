set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_properties_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
python tools/gpu_probe.py 120000 64,512 2>&1 | grep -v "^ns"
