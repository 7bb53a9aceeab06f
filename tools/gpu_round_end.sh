# Everything profiles/ of a round comes from, in one call (TAG=r03_x ...; PMC_ONLY=1: only the counter passes; SKIP_TESTS=1):
#   GPU tests; the default bench line; rocprofv3 kernel stats of the same command and of a one-context run; PMC passes (SQ,
#   SQ2, FETCH_SIZE, WRITE_SIZE in separate runs) for three shapes: 1024x436 at one and at eight pairs per launch, 1920x1080 and
#   3840x2160 at patch radius 17 (BASELINE configs[3], [4]); bench lines for those shapes.
set -x
R=$GRAFT_REPO_ROOT
T=${TAG:-r03}
cd $R && mkdir -p gpurun_out/$T
HD="--width 1920 --height 1080 --batch 1 --inflight 1"
UHD="--width 3840 --height 2160 --patch-r 17 --batch 1 --inflight 1"
if [ -z "$PMC_ONLY" ]; then
[ -n "$SKIP_TESTS" ] || timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/$T/gpu_tests.txt
fi
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$T
SQ="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
COMMON="--repeats 1 --no-cpu-baseline --no-extras"
# label | bench arguments
while IFS='|' read -r label bargs; do
  [ -z "$label" ] && continue
  rm -rf $O/pmc_sq_$label $O/pmc_sq2_$label $O/pmc_fetch_$label $O/pmc_write_$label
  rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$label -- python3 $R/bench.py $bargs $COMMON > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$label -- python3 $R/bench.py $bargs $COMMON > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$label -- python3 $R/bench.py $bargs $COMMON > /dev/null 2>&1
  case $label in single|batch8) rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_sq2_$label -- python3 $R/bench.py $bargs $COMMON > /dev/null 2>&1 ;; esac
done <<LIST
single|--steps 8 --warmup 4 --batch 1 --inflight 1
batch8|--steps 16 --warmup 8 --batch 8 --inflight 1
hd|$HD --steps 4 --warmup 2
uhd17|$UHD --steps 2 --warmup 1
LIST
# the constants the bench lines below report (valid for exactly these device sources), derived on the box from the passes above
cd $R && python tools/store_profiles.py $T > gpurun_out/$T/pmc_constants_summary.json 2> gpurun_out/$T/store_profiles.err
if [ -z "$PMC_ONLY" ]; then
python bench.py --verify-config3 > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err; cut -c1-220 gpurun_out/$T/bench_default.json
python bench.py --batch 1 --inflight 3 --no-cpu-baseline --no-extras > gpurun_out/$T/bench_streams3.json 2>/dev/null
python bench.py $HD --inflight 3 --steps 24 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/$T/bench_hd.json 2>/dev/null
python bench.py $UHD --steps 4 --warmup 1 --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/$T/bench_4k_r17.json 2>/dev/null
# the driver's N > 1 launch form, two ranks sharing this box's one GPU (RCCL refuses two ranks per device: the ranks agree on gloo)
EPPM_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --verify-config3 --no-cpu-baseline > gpurun_out/$T/bench_torchrun_2ranks_1gpu.json 2> gpurun_out/$T/bench_torchrun.err; cut -c1-160 gpurun_out/$T/bench_torchrun_2ranks_1gpu.json
fi
cd /tmp
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --steps 30 --warmup 3 --batch 1 --inflight 1 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batch8 -- python3 $R/bench.py --steps 32 --warmup 8 --batch 8 --inflight 1 $COMMON > /dev/null 2>&1
fi
ls $O
