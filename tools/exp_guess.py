import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from eppm_amd import synth
G = int(sys.argv[1])
a, b, _, _ = synth.make_pair(436, 1024, seed=1234)
e = eppm_amd.EPPM(params=eppm_amd.Params(num_guess=G))
e.init(a, b, 436, 1024)
for _ in range(4):
    e.set_data(a, b); e.compute_flow()
