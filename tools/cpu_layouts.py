"""GPU-box probe (CPU only): the whole-host rate of the CPU oracle by process layout -- bench.cpu_whole_host with one round each.
Why bench.py's `cpu_baseline.whole_host` stays inside the cgroup's CPU quota (16 CPUs on this pool) and uses one hardware thread per
physical core.  usage: cpu_layouts.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print("cgroup cpu quota (CPUs):", bench.cpu_quota(), flush=True)

W, H = 1024, 436
for name, kw in (("16 threads per process, physical cores only", dict(threads_per_proc=16)),
                 ("8 threads per process, physical cores only", dict(threads_per_proc=8)),
                 ("32 threads per process, physical cores only", dict(threads_per_proc=32)),
                 ("16 threads per process, physical cores only, OMP_WAIT_POLICY=passive", dict(threads_per_proc=16, env_extra={"OMP_WAIT_POLICY": "passive"})),
                 ("16 threads per process, every hardware thread, OMP_WAIT_POLICY=passive", dict(threads_per_proc=16, smt=True, env_extra={"OMP_WAIT_POLICY": "passive"}))):
    r = bench.cpu_whole_host(W, H, rounds=1, budget_s=1.0, ignore_quota=True, **kw)
    print(name, "->", json.dumps({k: r.get(k) for k in ("value", "processes", "cores", "round_s", "pair_s_min_max", "error")}), flush=True)
