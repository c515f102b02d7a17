# Round-end evidence in one gpurun call: GPU tests, bench (+ CPU legs), traced bench / counters / iteration profile,
# NDT / GICP kernel stats, probes.  Usage: bash tools/evidence_all.sh <tag>
set -u
tag=${1:-r02d}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/$tag/pytest_gpu.txt; tail -3 gpurun_out/$tag/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1; tail -1 gpurun_out/$tag/smoke.txt
bash tools/round_profile.sh $tag > gpurun_out/$tag/round_profile.log 2>&1; head -c 600 gpurun_out/$tag/bench_line.json; echo
cat gpurun_out/$tag/timed_region_trace_average.txt gpurun_out/$tag/iteration_profile.txt
bash tools/evidence_ndt_gicp.sh $tag > gpurun_out/$tag/ndt_gicp.log 2>&1; grep "ms/align" gpurun_out/$tag/ndt_gicp.log | cut -c1-110
python tools/single_pair_probe.py > gpurun_out/$tag/single_pair_probe.txt 2>&1; cat gpurun_out/$tag/single_pair_probe.txt
python tools/pm_probe.py > gpurun_out/$tag/pm_probe.txt 2>&1; cat gpurun_out/$tag/pm_probe.txt | cut -c1-140
python tools/gpu_probe.py 120000 1,16,64,512 > gpurun_out/$tag/gpu_probe.txt 2>&1; grep "align/s" gpurun_out/$tag/gpu_probe.txt | cut -c1-100
# the reference's own search (nn_mode NABO): the bench figure alone with the walked-query counts, and the counter passes of its kernels
python tools/nabo_probe.py nocert=0,1 > gpurun_out/$tag/nabo_probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/$tag/nabo_probe.txt | cut -c1-330
bash tools/nabo_pmc.sh $tag/nabo_pmc_cert nocert=0 > gpurun_out/$tag/nabo_pmc_certificates.txt 2>&1
bash tools/nabo_pmc.sh $tag/nabo_pmc_nocert nocert=1 > gpurun_out/$tag/nabo_pmc_every_query_walked.txt 2>&1; head -4 gpurun_out/$tag/nabo_pmc_every_query_walked.txt | cut -c1-300
# the sequence driver: kernel stats of a 513-scan drive
bash tools/sequence_profile.sh > gpurun_out/$tag/sequence_driver_kernel_stats.txt 2>&1; head -12 gpurun_out/$tag/sequence_driver_kernel_stats.txt | cut -c1-200
