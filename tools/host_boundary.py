"""PCIe-inclusive rates through the host-pointer boundary (eppm_set_images + eppm_compute: RGB->RGBA, H2D, the path, D2H, copy into
the caller's planes) at WxH (default 1024x436): synchronous on one context, and pipelined by ONE host thread over 2, 3 and 4
contexts (eppm_compute_begin / eppm_compute_end, eppm_amd.shard.run_pairs_pipelined), on 12 distinct synthetic pairs.
usage: host_boundary.py [--json] [W H]      (--json: one JSON line, what bench.py's host_boundary leg reports)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eppm_amd
from eppm_amd import synth, shard
args = [a for a in sys.argv[1:] if a != "--json"]
w, h = (int(args[0]), int(args[1])) if len(args) >= 2 else (1024, 436)
pairs = [synth.make_pair(h, w, seed=1234 + i)[:2] for i in range(12)]
e = eppm_amd.EPPM(); e.init(h, w)
for i in range(3):
    e.set_data(*pairs[i]); e.compute_flow()
n = 24
t = time.perf_counter()
for i in range(n):
    e.set_data(*pairs[i % 12]); e.compute_flow()
dt_sync = (time.perf_counter() - t) / n
e.close()
pipe = {}
for k in (2, 3, 4):
    engs = []
    for _ in range(k):
        g = eppm_amd.EPPM(); g.init(h, w); engs.append(g)
    work = pairs * 3
    shard.run_pairs_pipelined(engs, work, range(2 * k))
    t = time.perf_counter()
    shard.run_pairs_pipelined(engs, work, range(len(work)))
    pipe[k] = (time.perf_counter() - t) / len(work)
    for g in engs:
        g.close()
best = min(pipe, key=pipe.get)
if "--json" in sys.argv:
    print(json.dumps({"unit": "Mflow-vectors/s", "sync": w * h / dt_sync / 1e6, "sync_ms_per_pair": dt_sync * 1e3,
                      "pipelined": w * h / pipe[best] / 1e6, "pipelined_ms_per_pair": pipe[best] * 1e3, "contexts_in_flight": best,
                      "pipelined_ms_per_pair_by_contexts": {str(k): v * 1e3 for k, v in pipe.items()},
                      "note": "host RGB in, host u/v out; includes RGB->RGBA, H2D 2x3wh B, D2H 8wh B and the copy into the caller's planes; "
                              "one host thread, 12 distinct pairs; measured in a process of its own (tools/host_boundary.py)"}))
else:
    print(f"host boundary {w}x{h}: {dt_sync*1e3:.3f} ms/pair, {w*h/dt_sync/1e6:.1f} Mflow-vectors/s synchronous")
    for k, v in pipe.items():
        print(f"host boundary, {k} contexts pipelined: {v*1e3:.3f} ms/pair, {w*h/v/1e6:.1f} Mflow-vectors/s")
