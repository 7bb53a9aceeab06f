"""Oracle wall time by OpenMP thread count on this host (which count should tests / the CPU baseline use?): orc_threads.py [H W]..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from eppm_amd import synth
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "default threads", O.num_threads())
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(120, 160), (480, 640)]
for (h, w) in sizes:
    a, b, _, _ = synth.make_pair(h, w, seed=3)
    for n in (8, 16, 32, 64, 128, 256):
        if n > os.cpu_count():
            break
        O.set_num_threads(n)
        t = time.time(); O.compute_flow(a, b); print(h, w, n, round(time.time() - t, 2), flush=True)
