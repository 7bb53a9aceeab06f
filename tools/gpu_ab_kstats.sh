# per-kernel average durations of library variants (tools/build_variant.sh) under rocprofv3 --kernel-trace --stats:
# VARIANTS="a b" [BENCH_ARGS="--batch 4 --inflight 1"] [FILTER=k_pm]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cp $R/eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for v in $VARIANTS; do
  cp $R/gpurun_variants/$v/libeppm_hip.so $R/eppm_amd/lib/libeppm_hip.so
  rm -rf $O/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$v -- python3 $R/bench.py --steps 12 --warmup 4 ${BENCH_ARGS:---batch 4 --inflight 1} --no-cpu-baseline --no-extras > $O/ks_$v.log 2>&1
  echo "== $v"; python3 - "$O/ks_$v" "${FILTER:-k_}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"] and float(r["Percentage"]) > 0.4:
        print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
done
cp /tmp/libeppm_hip.orig.so $R/eppm_amd/lib/libeppm_hip.so
