"""By hand on a GPU box: python tests/parity_large.py -- HIP vs oracle, bit-exact, at the benchmark sizes
(1024x436 and 1920x1080 synthetic pairs; the oracle needs a many-core host to finish in seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eppm_amd
from eppm_amd import synth
from oracle import oracle as O

bad = 0
for (h, w, mf) in [(436, 1024, 20.0), (1080, 1920, 40.0)]:
    a, b, _, _ = synth.make_pair(h, w, seed=1234, max_flow=mf)
    e = eppm_amd.EPPM()
    e.init(a, b, h, w)
    u, v = e.compute_flow()
    t = time.time()
    ou, ov = O.compute_flow(a, b)
    same = np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(v.view(np.uint32), ov.view(np.uint32))
    print(f"{w}x{h}: {'bit-identical' if same else 'MISMATCH'} (oracle {time.time() - t:.1f} s, {O.num_threads()} threads)", flush=True)
    bad += not same
sys.exit(1 if bad else 0)
