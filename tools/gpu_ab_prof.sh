# per-kernel average durations (rocprofv3, single stream) for each variant in VARIANTS
R=$GRAFT_REPO_ROOT
cd $R; cp eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for v in $VARIANTS; do
  cp $R/gpurun_variants/$v/libeppm_hip.so $R/eppm_amd/lib/libeppm_hip.so
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > /tmp/prof_$v.log 2>&1
  f=$(ls -t /tmp/prof_$v/*/*kernel_stats.csv | head -1)
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
print("== $v")
for r in rows[:${TOPN:-10}]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%9.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
done
cp /tmp/libeppm_hip.orig.so $R/eppm_amd/lib/libeppm_hip.so
