# PMC passes that keep the PatchMatch kernels (k_pm_*): single stream (inflight 1) and three streams (inflight 3).
# Counters in their own runs (kernel-trace only), FETCH_SIZE and WRITE_SIZE in separate passes.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
SQ="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM"
for n in 1 3; do
  rm -rf $R/gpurun_out/pm_sq_$n
  rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pm_sq_$n -- python3 $R/bench.py --steps 6 --warmup 3 --inflight $n --no-cpu-baseline > $R/gpurun_out/pm_sq_$n.log 2>&1
  tail -1 $R/gpurun_out/pm_sq_$n.log | cut -c1-200
done
rm -rf $R/gpurun_out/pm_fetch $R/gpurun_out/pm_write $R/gpurun_out/pm_stats_1 $R/gpurun_out/pm_stats_3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pm_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pm_write -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
for n in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm_stats_$n -- python3 $R/bench.py --steps 30 --warmup 3 --inflight $n --no-cpu-baseline > $R/gpurun_out/pm_stats_$n.log 2>&1
done
ls $R/gpurun_out/pm_sq_1/* | head
