cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
VARIANTS="base u1 u5 u10" ROUNDS=2 INFLIGHT="1" STEPS=40 BENCH_ARGS="--no-extras" bash tools/gpu_ab_stage.sh
