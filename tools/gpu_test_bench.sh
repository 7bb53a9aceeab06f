set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err; tail -3 gpurun_out/bench_latest.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_latest.json'))
print('ms/step', d['ms_per_step'], 'Mvec/s', d['value'])
for k,v in d['stage_ms'].items(): print(f'  {k:18s} {v:8.3f}')
PY
