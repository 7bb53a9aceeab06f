#!/bin/bash
# tools/gpu.sh MODE -- every GPU-box job of this repository in one script (run as: gpurun -- 'VAR=.. bash tools/gpu.sh MODE';
# outputs under gpurun_out/, summaries copied to profiles/ by tools/store_profiles.py).  Library variants for the A/B modes are built
# in the container with tools/build_variant.sh NAME [flags] into gpurun_variants/NAME (git-ignored, shipped by gpurun) and swapped into
# eppm_amd/lib in turn, interleaved over ROUNDS (noise inside one box is ~0.5 %, box to box ~3-8 %: never compare across calls).
#
#   test        GPU suite (PYTEST_ARGS, default "-x -q")               bench       the default bench line -> gpurun_out/bench_latest.json
#   quick       GPU suite + a short bench with stage times             stats       rocprofv3 --kernel-trace --stats of BENCH_ARGS -> gpurun_out/stats_$TAG
#   ab          VARIANTS="a b": default bench per variant (BENCH_ARGS) abstage     VARIANTS: single-stream stage times (tools/stage_times.py; SIZE=WxH PATCH_R= BATCH=)
#   vtest       VARIANTS: the GPU suite (K="expr") on each variant's libraries
#   abk         VARIANTS: per-kernel averages under rocprofv3 (BENCH_ARGS, FILTER=k_pm)
#   pmc         PMC="counters.." in one pass of BENCH_ARGS -> gpurun_out/pmc_$TAG + per-kernel summary
#   tcp         TA/TCP/TD counters per kernel (is a gather kernel L1 bound?)
#   pmiter      every PatchMatch launch of one run in order, with durations (BENCH_ARGS)
#   clock       shader clock and power under load                      inflight    throughput vs contexts in flight
#   round       everything profiles/ of a round comes from (TAG=r04_x; PMC_ONLY=1, SKIP_TESTS=1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
MODE=$1
TAG=${TAG:-r06}
QUIET="--no-cpu-baseline --no-extras"
# a variant that is missing or misspelt must stop the job, not benchmark the previous library under the new label; the test library
# travels with the product library (the pytest process computes with it) and a variant without one is an error for the modes that need it.
# LIB=tol: the variants are builds of the tolerance library (tools/build_variant.sh with TOL=1), run with EPPM_HIP_VARIANT=tol
LIBS="libeppm_hip.so libeppm_hip_test.so"
if [ "$LIB" = tol ]; then LIBS="libeppm_hip_tol.so"; export EPPM_HIP_VARIANT=tol; fi
swap_in() { for l in $LIBS; do cp $R/gpurun_variants/$1/$l $R/eppm_amd/lib/$l || { echo "swap_in: variant '$1' has no $l" >&2; exit 1; }; done; }
# the saved original is this job's own (mktemp): two A/B jobs on one box never restore each other's variant
keep_orig() { ORIG=$(mktemp -d /tmp/eppm_orig.XXXXXX) || exit 1; for l in $LIBS; do cp $R/eppm_amd/lib/$l $ORIG/ || exit 1; done
              trap 'for l in $LIBS; do cp $ORIG/$l $R/eppm_amd/lib/; done; rm -rf $ORIG' EXIT; }
# rocprofv3 runs: the synthetic pairs come from the file cache (filled here, by plain child processes), so that the profiled bench.py
# starts no `python -m eppm_amd.synth` workers under the profiler's preload (each would add an output directory of its own)
warm_synth() { (cd $R && python3 - <<'PY'
from eppm_amd import synth
jobs = [(436, 1024, 1234 + i, 20.0) for i in range(64)] + [(1080, 1920, 1234, 40.0), (2160, 3840, 1234, 60.0)]
synth.make_pairs_parallel(jobs)
PY
) ; }
prof_env() { warm_synth; cd /tmp; export TMPDIR=/tmp; }
kstats() { python3 - "$1" "${2:-k_}" <<'PY'
import csv, glob, os, sys
# the bench process's file: the one with kernel rows (child processes that never launch a kernel leave empty or no stats files)
fs = [f for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv") if "k_" in open(f).read()]
f = max(fs, key=os.path.getsize)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"] and float(r["Percentage"]) > 0.3:
        print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
}
case $MODE in
test)
  cd $R && timeout ${TEST_TIMEOUT:-3000} python -m pytest tests -m gpu ${PYTEST_ARGS:--x -q} ${K:+-k "$K"} 2>&1 | tail -${TAIL:-15} | tee $O/gpu_tests_latest.txt ;;
bench)
  cd $R && python bench.py $BENCH_ARGS > $O/bench_latest.json 2> $O/bench_latest.err; tail -3 $O/bench_latest.err; cut -c1-400 $O/bench_latest.json ;;
quick)
  cd $R && timeout ${TEST_TIMEOUT:-3000} python -m pytest tests -m gpu ${PYTEST_ARGS:--x -q} 2>&1 | tail -15 | tee $O/gpu_tests_latest.txt
  python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-other-configs $BENCH_ARGS > $O/bench_latest.json 2> $O/bench_latest.err; tail -3 $O/bench_latest.err
  python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_latest.json'))
print('ms/step', d['ms_per_step'], 'Mvec/s', d['value'], 'latency', d.get('latency_ms_per_pair'), 'verified', d.get('timed_region_verified'))
for k, v in (d.get('stage_ms') or {}).items(): print(f'  {k:18s} {v:8.3f}')
PY
  ;;
stats)
  prof_env; rm -rf $O/stats_$TAG
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$TAG -- python3 $R/bench.py ${BENCH_ARGS:---steps 32 --warmup 8 --batch 8 --inflight 1} --repeats 1 $QUIET > $O/stats_$TAG.log 2>&1
  tail -1 $O/stats_$TAG.log | cut -c1-200; kstats $O/stats_$TAG "${FILTER:-k_}" ;;
ab)
  cd $R; keep_orig
  for r in $(seq 1 ${ROUNDS:-2}); do for v in $VARIANTS; do swap_in $v
    python bench.py $QUIET --steps ${STEPS:-96} ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],4), d['timed_region_verified']['ok'], '/', d['timed_region_verified']['of'])"
  done; done ;;
vtest)
  # parity of library variants: the GPU suite (K = pytest -k expression) with each variant's libraries swapped in
  cd $R; keep_orig
  for v in $VARIANTS; do swap_in $v; echo "== $v"; timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${K:+-k "$K"} 2>&1 | tail -3; done ;;
abstage)
  cd $R; keep_orig
  for r in $(seq 1 ${ROUNDS:-3}); do for v in $VARIANTS; do swap_in $v; python tools/stage_times.py $v $r; done; done ;;
abk)
  prof_env; keep_orig
  for v in $VARIANTS; do swap_in $v; rm -rf $O/ks_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$v -- python3 $R/bench.py --steps 16 --warmup 8 ${BENCH_ARGS:---batch 8 --inflight 1} --repeats 1 $QUIET > $O/ks_$v.log 2>&1
    echo "== $v"; kstats $O/ks_$v "${FILTER:-k_}"
  done ;;
pmiter)
  # every PatchMatch launch of one run in order (BENCH_ARGS: default 8 pairs per launch, one context)
  prof_env; rm -rf $O/pmiter_$TAG
  rocprofv3 --kernel-trace --output-format csv -d $O/pmiter_$TAG -- python3 $R/bench.py --steps 16 --warmup 8 ${BENCH_ARGS:---batch 8 --inflight 1} --repeats 1 $QUIET > $O/pmiter_$TAG.log 2>&1
  python3 $R/tools/pm_by_iteration.py $O/pmiter_$TAG | tee $O/pmiter_$TAG.txt ;;
pmc)
  prof_env; rm -rf $O/pmc_$TAG
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_$TAG -- python3 $R/bench.py ${BENCH_ARGS:---steps 16 --warmup 8 --batch 8 --inflight 1} --repeats 1 $QUIET > $O/pmc_$TAG.log 2>&1
  tail -1 $O/pmc_$TAG.log | cut -c1-200; python3 $R/tools/pmc_summary.py $O/pmc_$TAG ${FILTER} | tee $O/pmc_$TAG.csv | head -${HEAD:-40} ;;
tcp)
  prof_env; B=${BATCH:-8}
  P1="TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
  P2="TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
  i=1
  for P in "$P1" "$P2"; do rm -rf $O/pmc_tcp$i
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_tcp$i -- python3 $R/bench.py --steps 8 --warmup 4 --batch $B --inflight 1 --repeats 1 $QUIET $BENCH_ARGS > $O/pmc_tcp$i.log 2>&1
    python3 $R/tools/pmc_summary.py $O/pmc_tcp$i > $O/pmc_tcp${i}_b$B.csv; i=$((i+1)); done ;;
clock)
  cd $R; python bench.py --steps 8000 --warmup 3 $QUIET > /tmp/b.json 2>/dev/null & BP=$!
  for i in $(seq 1 45); do
    echo "t=$i $(rocm-smi --showclocks 2>&1 | grep -i 'sclk' | head -1 | sed 's/.*(//;s/).*//') $(rocm-smi --showpower 2>&1 | grep -i 'power' | head -1 | sed 's/.*: //')"
    sleep 1; kill -0 $BP 2>/dev/null || break; done
  wait $BP; cut -c1-160 /tmp/b.json ;;
inflight)
  cd $R; for S in ${INFLIGHT:-1 2 3 4 6}; do
    python bench.py --steps 96 --warmup 8 --inflight $S $QUIET $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight $S pairs in flight',d['config']['pairs_in_flight_per_gpu'],'ms/step %.3f'%d['ms_per_step'],'Mvec/s %.1f'%d['value'])"
  done ;;
round)
  # GPU tests; the default bench line; rocprofv3 kernel stats of the same command and of one-context runs; PMC passes (SQ, SQ2, FETCH_SIZE,
  # WRITE_SIZE in separate runs) for 1024x436 at one and at eight pairs per launch, 1920x1080 and 3840x2160 at patch radius 17.
  set -x
  T=$TAG; cd $R && mkdir -p $O/$T
  HD="--width 1920 --height 1080 --batch 1 --inflight 1"
  UHD="--width 3840 --height 2160 --patch-r 17 --batch 1 --inflight 1"
  if [ -z "$PMC_ONLY" ] && [ -z "$SKIP_TESTS" ]; then timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/$T/gpu_tests.txt; fi
  prof_env; OT=$O/$T
  SQ="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM"
  SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
  COMMON="--repeats 1 $QUIET"
  # the PMC passes for both libraries: exact (labels single, batch8, hd, uhd17) and tolerance (tol_*); separate --pmc runs, never with a trace domain
  for lib in exact tol; do
    pre=""; [ $lib = tol ] && pre="tol_"
  while IFS='|' read -r label bargs; do
    [ -z "$label" ] && continue
    label=$pre$label
    rm -rf $OT/pmc_sq_$label $OT/pmc_sq2_$label $OT/pmc_fetch_$label $OT/pmc_write_$label
    rocprofv3 --kernel-trace --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d $OT/pmc_sq_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1
    case $label in single|batch8|tol_single|tol_batch8)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OT/pmc_fetch_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OT/pmc_write_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OT/pmc_sq2_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1 ;;
    hd|uhd17)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OT/pmc_fetch_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OT/pmc_write_$label -- python3 $R/bench.py --library $lib $bargs $COMMON > /dev/null 2>&1 ;;
    esac
  done <<LIST
single|--steps 8 --warmup 4 --batch 1 --inflight 1
batch8|--steps 16 --warmup 8 --batch 8 --inflight 1
hd|$HD --steps 4 --warmup 2
uhd17|$UHD --steps 2 --warmup 1
LIST
  done
  # the constants the bench lines below report (valid for exactly these device sources), derived on the box from the passes above
  cd $R && python tools/store_profiles.py $T > $OT/pmc_constants_summary.json 2> $OT/store_profiles.err
  if [ -z "$PMC_ONLY" ]; then
    python bench.py > $OT/bench_default.json 2> $OT/bench_default.err; cut -c1-220 $OT/bench_default.json
    python bench.py --steps 20 --warmup 5 > $OT/bench_like_driver.json 2>/dev/null
    python bench.py --batch 1 --inflight 3 $QUIET > $OT/bench_streams3.json 2>/dev/null
    # the driver's N > 1 launch form, two ranks sharing this box's one GPU (RCCL refuses two ranks per device: the ranks agree on gloo)
    EPPM_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline > $OT/bench_torchrun_2ranks_1gpu.json 2> $OT/bench_torchrun.err; cut -c1-160 $OT/bench_torchrun_2ranks_1gpu.json
    prof_env
    rocprofv3 --kernel-trace --stats --output-format csv -d $OT/stats_default -- python3 $R/bench.py $QUIET > $OT/bench_under_rocprof.json 2>/dev/null
    rocprofv3 --kernel-trace --stats --output-format csv -d $OT/stats_single -- python3 $R/bench.py --steps 30 --warmup 3 --batch 1 --inflight 1 $COMMON > /dev/null 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OT/stats_batch8 -- python3 $R/bench.py --steps 32 --warmup 8 --batch 8 --inflight 1 $COMMON > /dev/null 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OT/stats_tol_batch8 -- python3 $R/bench.py --library tol --steps 32 --warmup 8 --batch 8 --inflight 1 $COMMON > /dev/null 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OT/stats_tol_default -- python3 $R/bench.py --library tol $QUIET > $OT/bench_tol_under_rocprof.json 2>/dev/null
    python3 $R/tools/tolerance_epe.py --all > $OT/tolerance_epe.json 2>/dev/null
  fi
  ls $OT ;;
*)
  sed -n 2,17p "$0"; exit 1 ;;
esac
