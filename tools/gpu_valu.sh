R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_valu -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
