"""Host-side cost of enqueueing one pair (set_data_device + compute_flow_device, ~150 kernel launches) with the queue
empty: if it approaches the GPU time per step, the issuing thread limits throughput with pairs in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eppm_amd
from eppm_amd import synth

h, w = 436, 1024
dev = torch.device("cuda:0")
a, b, _, _ = synth.make_pair(h, w, seed=1234)
def to_dev(img):
    rgba = np.zeros((h, w, 4), np.uint8); rgba[..., :3] = img
    return torch.from_numpy(rgba).to(dev)
A, Bm = to_dev(a), to_dev(b)
F = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
e = eppm_amd.EPPM(); e.init(h, w)
for _ in range(3):
    e.set_data_device(A.data_ptr(), Bm.data_ptr(), w * 4); e.compute_flow_device(F.data_ptr())
e.synchronize()
ts = []
for _ in range(20):
    e.synchronize()
    t0 = time.perf_counter()
    e.set_data_device(A.data_ptr(), Bm.data_ptr(), w * 4); e.compute_flow_device(F.data_ptr())
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
ts = np.array(ts) * 1e3
print("enqueue one pair: %.3f ms (median), until done: %.3f ms" % (np.median(ts[:, 0]), np.median(ts[:, 1])))
