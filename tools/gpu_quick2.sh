cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not config5" 2>&1 | tail -4
python tools/stage_times.py fixed 1
python bench.py --no-cpu-baseline > gpurun_out/bench_c1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_c1.json')); print('streams3', d['value'], d['ms_per_step'], 'lat', d['latency_ms_per_pair']); print(d['config3']); print(d['stage_ms']); print(d['host_boundary'], d['cold_ms'])"
for cfg in "4 3" "8 2" "2 3" "3 3"; do set -- $cfg; python bench.py --no-cpu-baseline --no-extras --batch $1 --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 inflight $2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
