cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for r in 1 2; do for v in $VARIANTS; do
  cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
  python bench.py --width 1920 --height 1080 --steps 12 --warmup 2 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$v HD ms/step %.2f blf %.3f'%(d['ms_per_step'], s['flow_blf_L0']+s['flow_blf_L1']+s['flow_blf_final']))"
done; done
cp /tmp/libeppm_hip.orig.so eppm_amd/lib/libeppm_hip.so
