# A/B timing on one box: the variants built by tools/build_variant.sh are swapped in turn into eppm_amd/lib and
# benchmarked, ROUNDS times interleaved (box-to-box and run-to-run noise is ~3 %).  VARIANTS="a b" ROUNDS=3
cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in $VARIANTS; do
    cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
    for S in ${INFLIGHT:-1 3}; do
      python bench.py --steps ${STEPS:-48} --warmup 6 --inflight $S --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$v round $r inflight $S ms/step %.3f'%d['ms_per_step'], 'lat %.3f'%d['latency_ms_per_pair'], 'pm %.3f post %.3f c2fL1 %.3f c2fL0 %.3f blf %.3f'%(s['patchmatch'],s['l2_post'],s.get('c2f_refine_L1',0),s.get('c2f_refine_L0',0),s.get('flow_blf_L0',0)+s.get('flow_blf_L1',0)+s.get('flow_blf_final',0)))"
    done
  done
done
cp /tmp/libeppm_hip.orig.so eppm_amd/lib/libeppm_hip.so
