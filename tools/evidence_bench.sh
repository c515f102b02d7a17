# the default bench line, then the same with the NdtWithGicp CPU leg (numpy oracle, ~100 s) to record that baseline
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
t0=$(date +%s)
python bench.py > gpurun_out/r02d/bench_line.json 2> gpurun_out/r02d/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench_line.json'))
print(d['value'], d['roofline']['frac'], d['figures']['identity_guess']['value'], d['figures']['early_exit']['value'])
print(d['single_pair'])
print({k:(v.get('value'), v.get('ms_per_alignment'), v.get('target_kept')) for k,v in d['other_workloads'].items()})
PY
SMHIP_BENCH_GICP_CPU=1 python bench.py --steps 2 --warmup 1 --no-figures > gpurun_out/r02d/bench_line_gicp_cpu.json 2> gpurun_out/r02d/bench2.err; echo "bench2 rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench_line_gicp_cpu.json'))
g=d['other_workloads']['ndt_gicp']
print(g.get('cpu_baseline'), g.get('parity'))
json.dump(g.get('cpu_baseline'), open('gpurun_out/r02d/r02_gicp_cpu_baseline.json','w'))
PY
