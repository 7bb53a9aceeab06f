"""EPE of the library selected by EPPM_HIP_VARIANT (default or approx) against the CPU oracle on the bundled frame10/frame11
pair: the check behind the opt-in approx-exp build (v_exp_f32 instead of the shared exp formula).  Prints one JSON line."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, eppm_amd
from oracle import oracle as O
from conftest import read_ppm, GOLDEN
a, b = read_ppm(os.path.join(GOLDEN, "frame10.ppm")), read_ppm(os.path.join(GOLDEN, "frame11.ppm"))
e = eppm_amd.EPPM(); e.init(a, b, 480, 640); u, v = e.compute_flow()
ou, ov = O.compute_flow(a, b)
epe = np.sqrt((u - ou) ** 2 + (v - ov) ** 2)
print(json.dumps({"library": eppm_amd.lib().eppm_version().decode(), "pair": "frame10/11 640x480", "epe_mean_px": float(epe.mean()),
                  "pixels_differing": float((epe > 0).mean()), "epe_max_px": float(epe.max())}))
