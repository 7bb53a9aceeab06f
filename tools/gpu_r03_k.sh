# evaluation cache of the sweeps: parity (every PatchMatch / pipeline test), A/B against no cache, single pair: classic + cache vs speculative + cache
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_k
timeout 2400 python -m pytest tests -m gpu -x -q -k "not approx and not two_ranks and not host_boundary" 2>&1 | tail -6 | tee gpurun_out/r03_k/tests.txt
VARIANTS="nocache cache cacheall" ROUNDS=2 bash tools/gpu_ab_stage.sh 2>&1 | grep -v "^+" | cut -c1-60 | tee gpurun_out/r03_k/ab_stage.txt
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for r in 1 2; do for v in nocache cache; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so; python bench.py --no-cpu-baseline --no-extras --steps 96 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"; done; done | tee gpurun_out/r03_k/ab_bench.txt
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_k
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --batch 8 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --batch 1 --inflight 1 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for d in stats_b8 stats_b1; do cp $(ls $O/$d/*/*kernel_trace.csv | head -1) $O/${d}_kernel_trace.csv; cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
