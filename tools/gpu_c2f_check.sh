set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "c2f or end_to_end or odd_size or r17 or R17 or generic or degenerate or full_640" 2>&1 | tail -4
timeout 600 python tests/parity_large.py 2>&1 | tail -4
for S in 1 3; do
python bench.py --steps 24 --warmup 3 --inflight $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('inflight $S ms/step %.3f'%d['ms_per_step'], 'lat %.3f'%d['latency_ms_per_pair'], 'pm %.3f post %.3f c2fL1 %.3f c2fL0 %.3f blf %.3f'%(s['patchmatch'],s['l2_post'],s['c2f_refine_L1'],s['c2f_refine_L0'],s['flow_blf_L0']+s['flow_blf_L1']+s['flow_blf_final']))"
done
