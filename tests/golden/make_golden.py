#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

Inputs: frame10.ppm / frame11.ppm in this directory -- the reference's bundled Middlebury pair, data
files copied from /root/reference (sha256 in MANIFEST.json).  Outputs:
  crop160_stages.npz   every intermediate plane of the whole path on the 160x120 centre crop
  pm_iters.npz         level-2 NNF + cost of the crop after 0, 1 and 10 PatchMatch iterations
  xorwow.json          first draws of the restated XORWOW streams (seed 1234, blocks 0,1,7,1000)
  MANIFEST.json        sha256 of the inputs and of the oracle's full 640x480 flow (u then v, float32 LE)
The reference ships no golden outputs: these pin the ORACLE against regressions and give
the GPU tests fixed expectations; parity with the CUDA original itself stays unpinned.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import read_ppm  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    a = read_ppm(os.path.join(HERE, "frame10.ppm"))
    b = read_ppm(os.path.join(HERE, "frame11.ppm"))
    ca, cb = a[180:300, 240:400].copy(), b[180:300, 240:400].copy()
    u, v, st = O.compute_flow(ca, cb, dump=True)
    keep = {k: val for k, val in st.items() if isinstance(val, np.ndarray)}
    keep["u"], keep["v"] = u, v
    np.savez_compressed(os.path.join(HERE, "crop160_stages.npz"), **keep)

    i1, i2, c1, c2 = st["img1_L2"], st["img2_L2"], st["cen1_L2"], st["cen2_L2"]
    pm = {}
    for it in (0, 1, 10):
        nnf, cost = O.patchmatch(i1, i2, c1, c2, iters_done=it)
        pm[f"nnf_it{it}"], pm[f"cost_it{it}"] = nnf, cost
    np.savez_compressed(os.path.join(HERE, "pm_iters.npz"), **pm)

    xw = {str(sub): [int(x) for x in O.xorwow_stream(1234, sub, 8)] for sub in (0, 1, 7, 1000)}
    json.dump(xw, open(os.path.join(HERE, "xorwow.json"), "w"), indent=1)

    fu, fv = O.compute_flow(a, b)
    man = {
        "frame10.ppm": hashlib.sha256(open(os.path.join(HERE, "frame10.ppm"), "rb").read()).hexdigest(),
        "frame11.ppm": hashlib.sha256(open(os.path.join(HERE, "frame11.ppm"), "rb").read()).hexdigest(),
        "oracle_flow_640x480_sha256": hashlib.sha256(fu.tobytes() + fv.tobytes()).hexdigest(),
        "oracle_flow_640x480_mean_u_v": [float(fu.mean()), float(fv.mean())],
    }
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)
    print(man)


if __name__ == "__main__":
    main()
