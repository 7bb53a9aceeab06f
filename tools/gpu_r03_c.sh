# round 3, third GPU call: phase A with the LDS target window -- parity, A/B against the gather form and the classic sweeps, and
# per-kernel durations (rocprofv3 kernel trace) at 4 pairs per launch and at one pair per launch
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_c
timeout 1500 python -m pytest tests -m gpu -x -q -k "substages or speculative or patchmatch_launcher or fuzz_parity_fixed or extreme" 2>&1 | tail -8 | tee gpurun_out/r03_c/tests.txt
VARIANTS="spec99 nowin win win2" ROUNDS=2 bash tools/gpu_ab_stage.sh 2>&1 | grep -v "^+" | cut -c1-60 | tee gpurun_out/r03_c/ab_stage.txt
VARIANTS="spec99 nowin win win2" bash tools/gpu_ab4.sh 2>&1 | grep -v "^+" | tee gpurun_out/r03_c/ab_bench.txt
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_c
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --batch 4 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --batch 1 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for d in stats_b4 stats_b1; do f=$(ls $O/$d/*/*kernel_stats.csv | head -1); echo $d; cut -d, -f1-4 $f | head -24; done
# keep only the small summaries
for d in stats_b4 stats_b1; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
