R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p4k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4k -- python3 $R/bench.py --width 3840 --height 2160 --patch-r 17 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/p4k/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
