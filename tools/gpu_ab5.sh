cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
cp gpurun_variants/p2/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "c2f or end_to_end or bundled" 2>&1 | tail -3
VARIANTS="p1 p2" ROUNDS=2 bash tools/gpu_ab_stage.sh
VARIANTS="p1 p2" bash tools/gpu_ab4.sh
