cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "patchmatch or end_to_end or jump or four" 2>&1 | tail -3
VARIANTS="cur pf" ROUNDS=3 bash tools/gpu_ab_stage.sh
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for r in 1 2; do for v in cur pf; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so; python bench.py --no-cpu-baseline --no-extras --steps 96 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v default bench', d['value'], d['ms_per_step'])"; done; done
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
