set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "c2f or end_to_end or bundled or odd_size or tiny or sintel" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "config2 or config4 or fuzz" 2>&1 | tail -5
VARIANTS="nowin win2" ROUNDS=3 INFLIGHT="1 3" STEPS=60 bash tools/gpu_ab.sh
cp gpurun_variants/approx/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
python tools/approx_exp_epe.py
python bench.py --steps 60 --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('approx_exp ms/step', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
