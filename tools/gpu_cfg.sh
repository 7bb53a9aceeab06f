cd $GRAFT_REPO_ROOT
for r in 1 2; do for cfg in "4 3" "4 4" "5 3" "6 3" "4 2" "6 2" "3 4" "2 4"; do set -- $cfg; python bench.py --no-cpu-baseline --no-extras --steps 120 --batch $1 --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 inflight $2', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
