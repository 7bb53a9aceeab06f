"""EPE of the library currently in eppm_amd/lib against the CPU oracle on the bundled frame10/frame11 pair and on the 1024x436
benchmark pair (used to judge the measurement-only -DEPPM_APPROX_EXP build: v_exp_f32 instead of the shared exp formula)."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from oracle import oracle as O
from eppm_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import read_ppm, GOLDEN
a, b = read_ppm(os.path.join(GOLDEN, "frame10.ppm")), read_ppm(os.path.join(GOLDEN, "frame11.ppm"))
e = eppm_amd.EPPM(); e.init(a, b, 480, 640); u, v = e.compute_flow()
ou, ov = O.compute_flow(a, b)
epe = np.sqrt((u - ou) ** 2 + (v - ov) ** 2)
print(json.dumps({"pair": "frame10/11 640x480", "epe_mean_px": float(epe.mean()), "pixels_differing": float((epe > 0).mean()), "epe_max_px": float(epe.max())}))
