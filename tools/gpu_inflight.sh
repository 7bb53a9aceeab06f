cd $GRAFT_REPO_ROOT
for S in 1 2 3 4 6; do
python bench.py --steps 48 --warmup 6 --inflight $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inflight',d['config']['pairs_in_flight_per_gpu'],'ms/step %.3f'%d['ms_per_step'],'Mvec/s %.1f'%d['value'], 'c2f_L0 %.3f'%d['stage_ms']['c2f_refine_L0'], 'pm %.3f'%d['stage_ms']['patchmatch'])"
done
