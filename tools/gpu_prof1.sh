# rocprofv3 kernel stats of a single-stream run (latency mode)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_lat
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat -- python3 $R/bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > $R/gpurun_out/prof_lat.log 2>&1
tail -1 $R/gpurun_out/prof_lat.log | cut -c1-200
