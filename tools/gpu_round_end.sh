# Everything profiles/ of a round comes from, in one call (TAG=r02_d ...; PMC_ONLY=1: only the counter passes, e.g. to refresh
# profiles/pmc_constants.json after an edit of the hashed kernel sources that does not change the kernels):
#   GPU tests; the default bench line; rocprofv3 kernel stats of the same command; PMC passes (SQ, FETCH_SIZE, WRITE_SIZE in
#   separate runs) for one pair per launch on one stream and for the default batch of 4 pairs per launch; HD and 4K bench lines.
set -x
R=$GRAFT_REPO_ROOT
T=${TAG:-r02}
cd $R && mkdir -p gpurun_out/$T
if [ -z "$PMC_ONLY" ]; then
[ -n "$SKIP_TESTS" ] || timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/$T/gpu_tests.txt
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err; cut -c1-220 gpurun_out/$T/bench_default.json
python bench.py --batch 1 --inflight 3 --no-cpu-baseline --no-extras > gpurun_out/$T/bench_streams3.json 2>/dev/null
python bench.py --width 1920 --height 1080 --steps 24 --warmup 3 --no-cpu-baseline --no-extras --batch 1 --inflight 3 > gpurun_out/$T/bench_hd.json 2>/dev/null
python bench.py --width 3840 --height 2160 --patch-r 17 --steps 4 --warmup 1 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > gpurun_out/$T/bench_4k_r17.json 2>/dev/null
# the driver's N > 1 launch form, two ranks sharing this box's one GPU (gloo for the barrier; RCCL refuses two ranks per device)
EPPM_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --dist-backend gloo --no-cpu-baseline > gpurun_out/$T/bench_torchrun_2ranks_1gpu.json 2> gpurun_out/$T/bench_torchrun.err; cut -c1-160 gpurun_out/$T/bench_torchrun_2ranks_1gpu.json
fi
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$T
SQ="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --steps 30 --warmup 3 --batch 1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
fi
for m in "1 single" "4 batch4"; do set -- $m
  rm -rf $O/pmc_sq_$2 $O/pmc_sq2_$2 $O/pmc_fetch_$2 $O/pmc_write_$2
  rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$2 -- python3 $R/bench.py --steps 8 --warmup 4 --batch $1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_sq2_$2 -- python3 $R/bench.py --steps 8 --warmup 4 --batch $1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$2 -- python3 $R/bench.py --steps 4 --warmup 4 --batch $1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$2 -- python3 $R/bench.py --steps 4 --warmup 4 --batch $1 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
ls $O
