set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_latest
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_latest -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_latest.log 2>&1
tail -1 $R/gpurun_out/prof_latest.log
