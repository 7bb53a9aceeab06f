# round 3, second GPU call: speculative sweeps -- parity tests, then A/B of the iteration from which the sweeps run speculatively
# (EPPM_SPEC_FROM_ITER = 0..4, 99 = never) by single-stream stage times and by the default bench
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_b
timeout 1500 python -m pytest tests -m gpu -x -q -k "substages or speculative or patchmatch_launcher or host_boundary or pinned or window or single_scale" 2>&1 | tail -15 | tee gpurun_out/r03_b/tests_new.txt
VARIANTS="spec99 spec0 spec1 spec2 spec3 spec4" ROUNDS=2 bash tools/gpu_ab_stage.sh 2>&1 | grep -v "^+" | tee gpurun_out/r03_b/ab_stage.txt
VARIANTS="spec99 spec1 spec2 spec3" bash tools/gpu_ab4.sh 2>&1 | grep -v "^+" | tee gpurun_out/r03_b/ab_bench.txt
