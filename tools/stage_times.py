"""Single-stream per-stage device times of the library in eppm_amd/lib at 1024x436 (A/B helper): stage_times.py LABEL ROUND"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from eppm_amd import synth
h, w = 436, 1024
a, b, _, _ = synth.make_pair(h, w, seed=1234)
e = eppm_amd.EPPM(); e.init(a, b, h, w)
for _ in range(3):
    e.compute_flow()
e.enable_stage_timing(1); e.stage_times(clear=True)
for _ in range(12):
    e.compute_flow()
agg = {}
for n, ms in e.stage_times(clear=True):
    agg.setdefault(n, []).append(ms)
print(" ".join(sys.argv[1:3]), " ".join(f"{k} {np.median(v):.4f}" for k, v in agg.items()), flush=True)
