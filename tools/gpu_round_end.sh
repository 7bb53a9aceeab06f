# everything the committed profiles/ of a round come from, in one call:
# GPU tests, default bench line, rocprofv3 kernel stats of the same command, PMC passes (VALU, FETCH_SIZE, WRITE_SIZE)
set -x
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py > gpurun_out/bench_default.json 2>/dev/null; cut -c1-220 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default $R/gpurun_out/pmc_valu $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_default.json 2> $R/gpurun_out/prof_default.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_valu -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
ls $R/gpurun_out/pmc_valu/* $R/gpurun_out/pmc_fetch/* $R/gpurun_out/pmc_write/* | head
