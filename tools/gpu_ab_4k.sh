cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for v in $VARIANTS; do
  cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
  python bench.py --width 3840 --height 2160 --patch-r 17 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$v 4K R17 ms/step %.1f pm %.1f'%(d['ms_per_step'], s['patchmatch']))"
done
cp /tmp/libeppm_hip.orig.so eppm_amd/lib/libeppm_hip.so
