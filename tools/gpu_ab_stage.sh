# A/B of library variants by single-stream stage times (tools/build_variant.sh): VARIANTS="a b" ROUNDS=3
cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/libeppm_hip.orig.so
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in $VARIANTS; do
    cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
    python tools/stage_times.py $v $r
  done
done
cp /tmp/libeppm_hip.orig.so eppm_amd/lib/libeppm_hip.so
