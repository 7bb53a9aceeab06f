set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "batch or config2" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err; tail -3 gpurun_out/bench_b1.err; cat gpurun_out/bench_b1.json
for cfg in "8 1" "8 2" "4 2" "4 3" "2 3"; do set -- $cfg; python bench.py --no-cpu-baseline --no-extras --batch $1 --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 inflight $2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
