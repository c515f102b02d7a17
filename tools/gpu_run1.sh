set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r02b/pytest_gpu.txt
tail -15 gpurun_out/r02b/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r02b/bench_line.json 2> gpurun_out/r02b/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b/bench_line.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('figures',{}).get('identity_guess',{}).get('value'), d.get('figures',{}).get('extrapolated_guess',{}).get('value'), d.get('figures',{}).get('early_exit',{}).get('value'))
print(d['kernel_ms_per_step'])
print({k:(v.get('value'), v.get('ms_per_alignment')) for k,v in d['other_workloads'].items()})
PY
