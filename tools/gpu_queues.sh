cd $GRAFT_REPO_ROOT
for Q in 4 8; do for S in 3 4 5 6; do
GPU_MAX_HW_QUEUES=$Q python bench.py --steps 48 --warmup 6 --inflight $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues $Q inflight',d['config']['pairs_in_flight_per_gpu'],'ms/step %.3f'%d['ms_per_step'],'Mvec/s %.1f'%d['value'])"
done; done
