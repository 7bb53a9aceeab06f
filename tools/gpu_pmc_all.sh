# PMC passes, single stream, one pair per launch (--batch 1 --inflight 1): SQ counters, FETCH_SIZE, WRITE_SIZE in separate runs;
# and the kernel stats of the default bench command.  TAG names the output directories under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-pmc}
mkdir -p $R/gpurun_out
SQ="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM"
rm -rf $R/gpurun_out/${T}_sq $R/gpurun_out/${T}_sq2 $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_sqb
rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq -- python3 $R/bench.py --steps 6 --warmup 3 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq2 -- python3 $R/bench.py --steps 6 --warmup 3 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sqb -- python3 $R/bench.py --steps 8 --warmup 4 --batch 4 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/bench.py --steps 3 --warmup 1 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_stats.json 2>/dev/null
cut -c1-200 $R/gpurun_out/${T}_stats.json
