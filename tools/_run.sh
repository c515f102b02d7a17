timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05o_pytest_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r05o_pytest_gpu.txt; grep -v amdgpu.ids gpurun_out/r05o_pytest_gpu.txt | tail -4
bash tools/r05_quick.sh r05o none "two:;one:no_overlap=1" "cv id"
