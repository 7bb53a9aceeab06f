"""PCIe-inclusive rates through the host-pointer boundary -- the window the reference's API itself names: set_data (RGB->RGBA,
H2D; driver .cpp:159-168) + compute_flow (the path, D2H, de-interleave; :217-306) -- at WxH (default 1024x436), 12 distinct pairs.

  sync            eppm_set_images + eppm_compute on ONE single-pair context, images and flow planes in memory registered with
                  eppm_host_register (the copy engine reads / writes the caller's memory; no host copy)
  sync_staged     the same on plain malloc'ed memory (one host copy each way through the context's pinned staging)
  sync_batch      eppm_batch_set_images + eppm_batch_compute, B pairs per call on one batch context, registered memory
  pipelined       ONE host thread keeps K batch contexts of B pairs in flight (eppm_batch_set_images,
                  eppm_batch_compute_begin_into / eppm_batch_compute_end), registered memory: the PCIe copies of one context
                  overlap the kernels of the others.  Default K x B = 3 x 8, the issue scheme of bench.py's `value`.
  pipelined_staged  the same on unregistered memory
  class           the C++ drop-in class (tools/runeppm --pairs: set_data + compute_flow on bao_alloc-shaped blocks), default
                  and with set_option("pin_caller_buffers", 1)

usage: host_boundary.py [--json] [--batch B] [--inflight K] [W H]      (--json: one JSON line, bench.py's host_boundary leg)"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import eppm_amd  # noqa: E402
from eppm_amd import synth  # noqa: E402


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    skip = set()
    for name in ("--batch", "--inflight"):
        if name in sys.argv:
            skip |= {sys.argv.index(name), sys.argv.index(name) + 1}
    pos = [a for i, a in enumerate(sys.argv[1:], 1) if i not in skip and not a.startswith("--")]
    w, h = (int(pos[0]), int(pos[1])) if len(pos) >= 2 else (1024, 436)
    B, K = opt("--batch", 8), opt("--inflight", 3)
    NP = 12
    plain = [p[:2] for p in synth.make_pairs_parallel([(h, w, 1234 + i, 20.0) for i in range(NP)])]
    pinned = []
    for a, b in plain:                                   # the same pairs in registered memory
        pa, pb = eppm_amd.pinned_empty((h, w, 3)), eppm_amd.pinned_empty((h, w, 3))
        pa[:], pb[:] = a, b
        pinned.append((pa, pb))

    def planes(n, reg):
        mk = (lambda: eppm_amd.pinned_empty((h, w), np.float32)) if reg else (lambda: np.empty((h, w), np.float32))
        return [(mk(), mk()) for _ in range(n)]

    def sync_single(pairs, reg, n=36):
        e = eppm_amd.EPPM()
        e.init(h, w)
        out = planes(1, reg)[0]
        for i in range(3):
            e.set_data(*pairs[i])
            e.compute_flow(out=out)
        t = time.perf_counter()
        for i in range(n):
            e.set_data(*pairs[i % NP])
            e.compute_flow(out=out)
        dt = (time.perf_counter() - t) / n
        e.close()
        return dt, out

    def sync_batch(pairs, nb, reps=6):
        e = eppm_amd.EPPMBatch(h, w, nb)
        out = planes(nb, True)
        groups = [[pairs[(g * nb + k) % NP] for k in range(nb)] for g in range(reps + 1)]
        e.set_data(groups[0])
        e.compute_flow(out=out)
        t = time.perf_counter()
        for g in range(1, reps + 1):
            e.set_data(groups[g])
            e.compute_flow(out=out)
        dt = (time.perf_counter() - t) / (reps * nb)
        e.close()
        return dt

    def pipelined(pairs, reg, nb, k, rounds=6):
        """one host thread, k batch contexts of nb pairs round robin: end(previous group of this context), set, begin"""
        engs = [eppm_amd.EPPMBatch(h, w, nb) for _ in range(k)]
        outs = [planes(nb, reg) for _ in range(k)]
        busy = [False] * k
        ngroups = rounds * k

        def run(first, count):
            for g in range(first, first + count):
                c = g % k
                if busy[c]:
                    engs[c].compute_flow_end(out=outs[c])
                engs[c].set_data([pairs[(g * nb + j) % NP] for j in range(nb)])
                engs[c].compute_flow_begin(out=outs[c])
                busy[c] = True
            for c in range(k):
                if busy[c]:
                    engs[c].compute_flow_end(out=outs[c])
                    busy[c] = False
        run(0, k)
        t = time.perf_counter()
        run(0, ngroups)
        dt = (time.perf_counter() - t) / (ngroups * nb)
        for e in engs:
            e.close()
        return dt

    def class_cli(pin, n=40):
        exe = os.path.join(ROOT, "eppm_amd", "lib", "runeppm")
        try:
            r = subprocess.run([exe, "--size", f"{w}x{h}", "--pairs", str(n), "--out", "/tmp/host_boundary_class.flo"] + (["--pin"] if pin else []),
                               capture_output=True, text=True, timeout=300)
            m = re.search(r"init hoisted\): ([0-9.]+) Mflow-vectors/s", r.stdout)
            return float(m.group(1)) if (r.returncode == 0 and m) else None
        except Exception:
            return None

    dt_sync, flow_reg = sync_single(pinned, True)
    dt_staged, flow_staged = sync_single(plain, False)
    assert np.array_equal(flow_reg[0], flow_staged[0]) and np.array_equal(flow_reg[1], flow_staged[1]), "registered and staged flows differ"
    dt_sb = sync_batch(pinned, B)
    dt_pipe = pipelined(pinned, True, B, K)
    dt_pipe_staged = pipelined(plain, False, B, K)
    dt_pipe1 = pipelined(pinned, True, 1, K)
    cls, cls_pin = class_cli(False), class_cli(True)
    rate = lambda dt: w * h / dt / 1e6  # noqa: E731
    res = {"unit": "Mflow-vectors/s",
           "sync": rate(dt_sync), "sync_ms_per_pair": dt_sync * 1e3,
           "sync_staged": rate(dt_staged), "sync_staged_ms_per_pair": dt_staged * 1e3,
           "sync_batch": rate(dt_sb), "sync_batch_ms_per_pair": dt_sb * 1e3, "sync_batch_pairs_per_call": B,
           "pipelined": rate(dt_pipe), "pipelined_ms_per_pair": dt_pipe * 1e3, "pairs_per_launch": B, "contexts_in_flight": K,
           "pipelined_staged": rate(dt_pipe_staged), "pipelined_single_pair_contexts": rate(dt_pipe1),
           "class_sync": cls, "class_sync_pinned": cls_pin,
           "note": "host RGB in, host u/v out, every pair: RGB->RGBA, H2D 2x3wh B, the path, de-interleave, D2H 8wh B.  sync / sync_batch / pipelined: "
                   "caller memory registered once with eppm_host_register (DMA from / into it, no host copy); *_staged: plain memory through the "
                   "context's pinned staging; class_*: the C++ drop-in class in steady state (runeppm --pairs), default and with pin_caller_buffers; "
                   "one host thread, 12 distinct pairs, a process of its own (tools/host_boundary.py)"}
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for k_, v in res.items():
            if k_ != "note":
                print(f"{k_:34s} {v}")


if __name__ == "__main__":
    main()
