set -u
timeout 1200 python -m pytest tests/test_ndt_gicp_gpu.py tests/test_ndt_gpu.py tests/test_target_cache_gpu.py tests/test_icp_gpu.py -m gpu -x -q 2>&1 | tail -4
python tools/gicp_probe.py 2>&1 | head -1 | cut -c1-100
python tools/ndt_probe.py 2>&1 | cut -c1-100
