# A/B of library variants on the 3840x2160 / patch radius 17 pair (BASELINE configs[4]): VARIANTS="a b"
cd $GRAFT_REPO_ROOT
cp eppm_amd/lib/libeppm_hip.so /tmp/orig.so
for r in 1 2; do for v in $VARIANTS; do cp gpurun_variants/$v/libeppm_hip.so eppm_amd/lib/libeppm_hip.so
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, '.')
import eppm_amd
from eppm_amd import synth
h, w = 2160, 3840
a, b, _, _ = synth.make_pair(h, w, seed=1234, max_flow=60.0) if $r == 1 or True else None
e = eppm_amd.EPPM(params=eppm_amd.Params(patch_r=17)); e.init(a, b, h, w)
e.compute_flow()
e.enable_stage_timing(1); e.stage_times(clear=True)
for _ in range(3): e.compute_flow()
agg = {}
for n, ms in e.stage_times(clear=True): agg.setdefault(n, []).append(ms)
print("$v", " ".join(f"{k} {np.median(v):.2f}" for k, v in agg.items()), "total", round(sum(np.median(v) for v in agg.values()), 1), flush=True)
PY
done; done
cp /tmp/orig.so eppm_amd/lib/libeppm_hip.so
