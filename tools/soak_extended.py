"""Extended race hunt on the GPU box (outside the suite): three 8-pair batch contexts kept in flight through the registered-memory host
boundary (eppm_batch_set_images / eppm_batch_compute_begin_into / _end), the issue scheme of bench.py's `value`, for N rounds at 1024x436
over 16 distinct pairs (seeds 1234 .. 1249 of BASELINE configs[2]); every flow of every round must hash like the committed oracle
flow of its pair (tests/golden/MANIFEST_config3.json).  usage: soak_extended.py [N]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import eppm_amd  # noqa: E402
from eppm_amd import synth  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST_config3.json")))
    h, w = man["h"], man["w"]
    NP, NB, K = 16, 8, 3
    pairs, want = [], []
    for i in range(NP):
        a, b, _, _ = synth.make_pair_cached(h, w, seed=man["seed0"] + i)
        pa, pb = eppm_amd.pinned_empty((h, w, 3)), eppm_amd.pinned_empty((h, w, 3))
        pa[:], pb[:] = a, b
        pairs.append((pa, pb))
        want.append(man["pairs"][str(i)]["flow_sha256"])
    engs = [eppm_amd.EPPMBatch(h, w, NB) for _ in range(K)]
    outs = [[(eppm_amd.pinned_empty((h, w), np.float32), eppm_amd.pinned_empty((h, w), np.float32)) for _ in range(NB)] for _ in range(K)]
    busy = [None] * K
    bad = checked = 0
    t0 = time.time()

    def collect(c):
        nonlocal bad, checked
        got = engs[c].compute_flow_end(out=outs[c])
        for idx, (u, v) in zip(busy[c], got):
            checked += 1
            if hashlib.sha256(u.tobytes() + v.tobytes()).hexdigest() != want[idx]:
                bad += 1
                print(f"MISMATCH pair {idx} (context {c})", flush=True)
        busy[c] = None
    for g in range(rounds * K):
        c = g % K
        if busy[c] is not None:
            collect(c)
        idx = [(g * 5 + 3 * j) % NP for j in range(NB)]          # a different mix of pairs per group and slot
        engs[c].set_data([pairs[i] for i in idx])
        engs[c].compute_flow_begin(out=outs[c])
        busy[c] = idx
    for c in range(K):
        if busy[c] is not None:
            collect(c)
    dt = time.time() - t0
    print(f"soak_extended: {checked} flows in {dt:.1f} s ({checked * w * h / dt / 1e6:.1f} Mflow-vectors/s incl. hashing), {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
