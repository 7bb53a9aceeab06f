#!/usr/bin/env python3
"""Golden hashes of ALL 64 pairs of BASELINE.json configs[2] (64 independent 1024x436 pairs, seeds 1234 .. 1297, pair i ->
rank i mod N: eppm_amd/shard.py), so that a multi-GPU run can verify every flow it produces whatever share a rank gets
(`bench.py --verify-config3`).  Run ONCE in the build container (about 20 minutes on 6 cores):

    python tests/golden/make_golden_config3.py

Hashes only (MANIFEST_config3.json: sha256 of the two input images, sha256 of u || v as float32 LE row-major, mean u, v);
the first eight pairs are also in MANIFEST_large.json with per-band hashes and crops (make_golden_large.py) and must agree.
The flows come from the CPU oracle (oracle/): regression pins of the oracle's reading, not pins against the CUDA binary.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from eppm_amd import synth  # noqa: E402

N_PAIRS, SEED0, H, W = 64, 1234, 436, 1024


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from oracle import oracle as O
    O.set_num_threads(int(os.environ.get("ORACLE_THREADS", "6")))
    mpath = os.path.join(HERE, "MANIFEST_config3.json")
    man = json.load(open(mpath)) if os.path.exists(mpath) else {"h": H, "w": W, "seed0": SEED0, "n_pairs": N_PAIRS, "pairs": {}}
    large = json.load(open(os.path.join(HERE, "MANIFEST_large.json")))
    for i in range(N_PAIRS):
        if str(i) in man["pairs"]:
            continue
        a, b, _, _ = synth.make_pair(H, W, seed=SEED0 + i)
        t = time.time()
        u, v = O.compute_flow(a, b)
        rec = {"seed": SEED0 + i, "img1_sha256": sha(a), "img2_sha256": sha(b),
               "flow_sha256": hashlib.sha256(u.tobytes() + v.tobytes()).hexdigest(),
               "mean_u_v": [float(u.mean(dtype=np.float64)), float(v.mean(dtype=np.float64))]}
        ref = large.get(f"sintel_{SEED0 + i}")
        if ref is not None and ref["flow_sha256"] != rec["flow_sha256"]:
            raise SystemExit(f"pair {i}: differs from MANIFEST_large.json")
        man["pairs"][str(i)] = rec
        json.dump(man, open(mpath, "w"), indent=1)
        print(i, rec["flow_sha256"][:16], round(time.time() - t, 1), "s", flush=True)


if __name__ == "__main__":
    main()
