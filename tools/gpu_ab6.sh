cd $GRAFT_REPO_ROOT
cp gpurun_variants/lpc8/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "patchmatch or end_to_end" 2>&1 | tail -3
VARIANTS="cur lpc8" bash tools/gpu_ab4.sh
VARIANTS="cur lpc8" ROUNDS=2 bash tools/gpu_ab_stage.sh
