# phase B at 4 lanes per chain: parity, A/B against 16 lanes, and against the classic form for one pair per launch; kernel stats
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_j
timeout 1500 python -m pytest tests -m gpu -x -q -k "substages or speculative or patchmatch or fuzz_parity_fixed or extreme or config2_sintel or config3_eight or batch_context_small" 2>&1 | tail -6 | tee gpurun_out/r03_j/tests.txt
VARIANTS="b16 b4 b4all" ROUNDS=2 bash tools/gpu_ab_stage.sh 2>&1 | grep -v "^+" | cut -c1-60 | tee gpurun_out/r03_j/ab_stage.txt
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for r in 1 2; do for v in b16 b4; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so; python bench.py --no-cpu-baseline --no-extras --steps 96 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"; done; done | tee gpurun_out/r03_j/ab_bench.txt
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_j
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --batch 8 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(ls $O/stats_b8/*/*kernel_stats.csv | head -1) $O/stats_b8_kernel_stats.csv; rm -rf $O/stats_b8
