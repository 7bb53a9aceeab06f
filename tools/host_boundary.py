"""PCIe-inclusive rate through the host-pointer boundary (eppm_set_images + eppm_compute) at 1024x436."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from eppm_amd import synth
h, w = 436, 1024
a, b, _, _ = synth.make_pair(h, w, seed=1234)
e = eppm_amd.EPPM(); e.init(h, w)
for _ in range(3):
    e.set_data(a, b); e.compute_flow()
t = time.perf_counter(); n = 20
for _ in range(n):
    e.set_data(a, b); e.compute_flow()
dt = (time.perf_counter() - t) / n
print(f"host boundary: {dt*1e3:.3f} ms/pair, {w*h/dt/1e6:.1f} Mflow-vectors/s (H2D 2.7 MB + D2H 3.6 MB per pair, synchronous)")

# the same with several contexts in flight on one host thread (eppm_compute_begin / eppm_compute_end)
from eppm_amd import shard
for k in (2, 3, 4):
    engs = []
    for _ in range(k):
        g = eppm_amd.EPPM(); g.init(h, w); engs.append(g)
    pairs = [(a, b)] * 24
    shard.run_pairs_pipelined(engs, pairs, range(6))
    t = time.perf_counter()
    out = shard.run_pairs_pipelined(engs, pairs, range(len(pairs)))
    dt = (time.perf_counter() - t) / len(pairs)
    print(f"host boundary, {k} contexts pipelined: {dt*1e3:.3f} ms/pair, {w*h/dt/1e6:.1f} Mflow-vectors/s")
    for g in engs:
        g.close()
