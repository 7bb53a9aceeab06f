R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats1 -- python3 $R/bench.py --steps 30 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras > $R/gpurun_out/stats1.log 2>&1
tail -1 $R/gpurun_out/stats1.log | cut -c1-300
