# the default bench line and the rocprofv3 kernel stats of the same command
set -x
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/bench_default.json 2>/dev/null; cat gpurun_out/bench_default.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_default.json 2> $R/gpurun_out/prof_default.err
tail -1 $R/gpurun_out/prof_default.err
