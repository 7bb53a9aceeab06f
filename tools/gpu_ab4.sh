cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for r in 1 2; do for v in $VARIANTS; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so; python bench.py --no-cpu-baseline --no-extras --steps 96 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"; done; done
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
