# round 3, fourth GPU call: packed parity target planes -- parity of every PatchMatch test, A/B (packed vs float4 gathers, phase-B
# register budget, speculative from iteration 2), per-kernel durations
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_d
timeout 1500 python -m pytest tests -m gpu -x -q -k "div_const or substages or speculative or patchmatch or fuzz_parity_fixed or extreme or radius_17 or config5 or config2_sintel or batch_context_small" 2>&1 | tail -8 | tee gpurun_out/r03_d/tests.txt
VARIANTS="spec99 nopack base gb7 spec2" ROUNDS=2 bash tools/gpu_ab_stage.sh 2>&1 | grep -v "^+" | cut -c1-60 | tee gpurun_out/r03_d/ab_stage.txt
VARIANTS="spec99 nopack base gb7 spec2" bash tools/gpu_ab4.sh 2>&1 | grep -v "^+" | tee gpurun_out/r03_d/ab_bench.txt
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_d
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --batch 4 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --batch 1 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for d in stats_b4 stats_b1; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
