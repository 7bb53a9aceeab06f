R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for G in 1 2 3 6 8; do
rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python3 $R/tools/exp_guess.py $G > /dev/null 2>&1
f=$(ls /tmp/pg/*/*kernel_stats.csv | head -1)
python3 - "$f" $G <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'random_search' in r['Name']: print('G=%s search avg_us %.1f min %.1f calls %s' % (sys.argv[2], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls']))
PY
done
