R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/phd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/phd -- python3 $R/bench.py --width 1920 --height 1080 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/phd/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
n=[int(r['Calls']) for r in rows if 'c2f_refine_tiled' in r['Name']][0]/2
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} per_pair_ms={float(r['TotalDurationNs'])/1e6/n:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
