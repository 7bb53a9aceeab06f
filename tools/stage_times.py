"""Single-stream per-stage device times of the library in eppm_amd/lib (A/B helper): stage_times.py LABEL ROUND
env: SIZE=WxH (default 1024x436), PATCH_R= patch radius (default 9), BATCH= pairs per launch (default 1: a single-pair context)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from eppm_amd import synth
w, h = (int(x) for x in os.environ.get("SIZE", "1024x436").split("x"))
R, NB = int(os.environ.get("PATCH_R", "9")), int(os.environ.get("BATCH", "1"))
mf = 20.0 if w <= 1024 else (40.0 if w <= 1920 else 60.0)
prm = eppm_amd.Params(patch_r=R)
if NB == 1:
    a, b, _, _ = synth.make_pair_cached(h, w, seed=1234, max_flow=mf)
    e = eppm_amd.EPPM(params=prm); e.init(a, b, h, w)
    run = e.compute_flow
else:
    pairs = [p[:2] for p in synth.make_pairs_parallel([(h, w, 1234 + i, mf) for i in range(NB)])]
    e = eppm_amd.EPPMBatch(h, w, NB, params=prm); e.set_data(pairs)
    run = e.compute_flow
n = 12 if w * h < 4e6 else 3
for _ in range(3 if n > 3 else 1):
    run()
e.enable_stage_timing(1); e.stage_times(clear=True)
for _ in range(n):
    run()
agg = {}
for name, ms in e.stage_times(clear=True):
    agg.setdefault(name, []).append(ms)
tot = sum(np.median(v) for k, v in agg.items() if k != "prepare")
print(" ".join(sys.argv[1:3]), " ".join(f"{k} {np.median(v) / NB:.4f}" for k, v in agg.items()), f"sum {tot / NB:.4f}", flush=True)
