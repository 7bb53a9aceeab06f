# first bench + rocprof on the GPU box
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*stats*" | head
