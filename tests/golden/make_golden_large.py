#!/usr/bin/env python3
"""Golden vectors of the BASELINE.json configurations that are too large for the oracle to re-run inside a GPU test
(run ONCE in the build container, about an hour on 8 cores):

    python tests/golden/make_golden_large.py [--only NAME ...]

For every case the CPU oracle computes the full flow of the synthetic pair (eppm_amd/synth.py, deterministic in the seed)
and MANIFEST_large.json keeps: sha256 of the two input images (so a test can tell "inputs differ on this host" from "flow
differs"), sha256 of u || v (float32 LE, row-major), the per-16-row-band sha256 list (localises a mismatch), the mean of u
and v, and large_crops.npz a 64x64 crop of u and v around the image centre.
Natural pairs (round 6): `bundled_640x480` = the reference's frame10/frame11 (BASELINE configs[0], the default three levels) and
`natural_1024x436` = the same frames scaled x1.6 and centre-cropped to the Sintel shape (eppm_amd/synth.py: natural_pair).
Cases: BASELINE configs[1] = `sintel_1234`; configs[2] = `sintel_1234 .. sintel_1241` (the 8 pairs one GPU of the 8-GPU
batch processes); configs[3] = `hd_1234`; configs[4] = `uhd_r17_1234` (3840x2160, patch radius 17).
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

CASES = [dict(name=f"sintel_{s}", h=436, w=1024, seed=s, max_flow=20.0, patch_r=9) for s in range(1234, 1242)]
CASES += [dict(name="hd_1234", h=1080, w=1920, seed=1234, max_flow=40.0, patch_r=9),
          dict(name="uhd_r17_1234", h=2160, w=3840, seed=1234, max_flow=60.0, patch_r=17)]
NATURAL = [dict(name="bundled_640x480", h=480, w=640, patch_r=9, source="tests/golden/frame10.ppm, frame11.ppm"),
           dict(name="natural_1024x436", h=436, w=1024, patch_r=9, source="synth.natural_pair(): the bundled frames x1.6 (bilinear, float64), rows 166..601")]
BAND = 16


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def flow_record(u, v):
    h, w = u.shape
    cy, cx = h // 2 - 32, w // 2 - 32
    rec = {"flow_sha256": hashlib.sha256(u.tobytes() + v.tobytes()).hexdigest(),
           "band_sha256": [hashlib.sha256(u[y:y + BAND].tobytes() + v[y:y + BAND].tobytes()).hexdigest()[:16] for y in range(0, h, BAND)],
           "mean_u_v": [float(u.mean(dtype=np.float64)), float(v.mean(dtype=np.float64))], "crop_origin_yx": [cy, cx]}
    return rec, u[cy:cy + 64, cx:cx + 64].copy(), v[cy:cy + 64, cx:cx + 64].copy()


def main():
    from oracle import oracle as O
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    mpath, cpath = os.path.join(HERE, "MANIFEST_large.json"), os.path.join(HERE, "large_crops.npz")
    man = json.load(open(mpath)) if os.path.exists(mpath) else {}
    crops = dict(np.load(cpath)) if os.path.exists(cpath) else {}
    for c in CASES + NATURAL:
        if only and c["name"] not in only:
            continue
        if "seed" in c:
            a, b, _, _ = synth.make_pair(c["h"], c["w"], seed=c["seed"], max_flow=c["max_flow"])
        else:
            a, b = synth.bundled_pair() if c["name"].startswith("bundled") else synth.natural_pair(c["h"], c["w"])
            assert a.shape == (c["h"], c["w"], 3)
        t = time.time()
        u, v = O.compute_flow(a, b, O.default_params(patch_r=c["patch_r"]))
        rec, cu, cv = flow_record(u, v)
        rec.update({k: c[k] for k in ("h", "w", "seed", "max_flow", "patch_r", "source") if k in c})
        rec.update({"img1_sha256": sha(a), "img2_sha256": sha(b), "oracle_seconds": round(time.time() - t, 1), "oracle_threads": O.num_threads()})
        man[c["name"]] = rec
        crops[c["name"] + "_u"], crops[c["name"] + "_v"] = cu, cv
        json.dump(man, open(mpath, "w"), indent=1)
        np.savez_compressed(cpath, **crops)
        print(c["name"], rec["flow_sha256"], rec["oracle_seconds"], "s", flush=True)


if __name__ == "__main__":
    main()
