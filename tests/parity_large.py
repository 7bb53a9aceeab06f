"""By hand on a GPU box: python tests/parity_large.py -- HIP vs oracle, bit-exact, at the benchmark sizes
(1024x436 and 1920x1080 synthetic pairs; the oracle needs a many-core host to finish in seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eppm_amd
from eppm_amd import synth
from oracle import oracle as O

bad = 0
cases = [(436, 1024, 20.0, 9), (1080, 1920, 40.0, 9)]
if "--4k" in sys.argv:                      # BASELINE.json configs[4]: 3840x2160, patch radius 17 (oracle: minutes on 128 cores)
    cases = [(2160, 3840, 60.0, 17)]
for (h, w, mf, R) in cases:
    a, b, _, _ = synth.make_pair(h, w, seed=1234, max_flow=mf)
    e = eppm_amd.EPPM(params=eppm_amd.Params(patch_r=R))
    e.init(a, b, h, w)
    u, v = e.compute_flow()
    t = time.time()
    ou, ov = O.compute_flow(a, b, O.default_params(patch_r=R))
    same = np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(v.view(np.uint32), ov.view(np.uint32))
    print(f"{w}x{h} R={R}: {'bit-identical' if same else 'MISMATCH'} (oracle {time.time() - t:.1f} s, {O.num_threads()} threads)", flush=True)
    bad += not same
sys.exit(1 if bad else 0)
