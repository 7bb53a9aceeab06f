cd $GRAFT_REPO_ROOT
for S in 1 3; do
python bench.py --steps 24 --warmup 3 --inflight $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('inflight $S ms/step %.3f'%d['ms_per_step'], 'pm %.3f post %.3f c2fL1 %.3f c2fL0 %.3f blf %.3f'%(s['patchmatch'],s['l2_post'],s['c2f_refine_L1'],s['c2f_refine_L0'],s['flow_blf_L0']+s['flow_blf_L1']+s['flow_blf_final']))"
done
