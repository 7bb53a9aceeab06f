# HBM-side traffic of the dominant kernel (k_c2f_refine_tiled): two separate --pmc passes as the
# MI355X guide prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: they do not fit one pass).
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
tail -1 $R/gpurun_out/pmc_write.log
