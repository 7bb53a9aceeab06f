#!/usr/bin/env python3
"""How far do the LEGAL alternative readings of the reference move the flow?  (CPU only; test infrastructure.)

The parity statement of this repository is "bit-identical to the lockstep oracle; parity against the CUDA binary unpinned":
the reference updates NNF / flow planes in place while other threads read them, draws from cuRAND and uses the SFU's __expf,
so a real CUDA run is ONE of many possible outputs.  This script runs the oracle under the other readings it can express
(oracle/eppm_oracle.c: orc_set_variant) and reports the end-point error of each against the lockstep oracle:

  sweep_serial     the segments of a line run one after the other in sweep direction with live seeds (propagation along the line)
  sweep_pixelL     lockstep seeds, but the doubly visited pixel L is reached by segment 0 before segment 1
  outlier_inplace, wmf_inplace, fill_inplace, smoothing_inplace
                   that stage reads the buffer it writes, in raster order (the far end of what a grid can do; degenerate for the
                   outlier vote, where every invalidated pixel stops supporting its neighbours)
  libm_expf        libm expf instead of the shared 2-ulp __expf formula
  other_stream     another seed scrambling (a different, equally plausible XORWOW stream)
  all_but_outlier  all of the above together except the degenerate raster-order outlier vote

on the bundled Middlebury pair (640x480) and on the 1024x436 synthetic pair of BASELINE configs[1].  Results go to
profiles/parity_envelope.json (committed; DESIGN.md section 3.6 quotes them); tests/test_oracle_cpu.py pins the small case.

usage: parity_envelope.py [--small]        (--small: 160x120 crop of the bundled pair only, a few seconds)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = [("sweep_serial", dict(sweep_order=1)), ("sweep_pixelL", dict(sweep_order=2)),
            ("outlier_inplace", dict(post_inplace=1)), ("wmf_inplace", dict(post_inplace=2)), ("fill_inplace", dict(post_inplace=4)),
            ("smoothing_inplace", dict(post_inplace=8)), ("libm_expf", dict(exp_mode=1)), ("other_stream", dict(seed_variant=1)),
            ("all_but_outlier", dict(sweep_order=1, post_inplace=14, exp_mode=1, seed_variant=1))]


def epe_stats(u, v, u0, v0):
    e = np.sqrt((u.astype(np.float64) - u0) ** 2 + (v.astype(np.float64) - v0) ** 2)
    return {"mean_epe_px": float(e.mean()), "p95_epe_px": float(np.percentile(e, 95)), "p99_epe_px": float(np.percentile(e, 99)),
            "max_epe_px": float(e.max()), "frac_differing": float((e > 0).mean()), "frac_over_1px": float((e > 1.0).mean())}


def envelope(a, b, O):
    O.set_variant()
    u0, v0 = O.compute_flow(a, b)
    out = {}
    for name, kw in VARIANTS:
        O.set_variant(**kw)
        t = time.time()
        try:
            u, v = O.compute_flow(a, b)
        finally:
            O.set_variant()
        out[name] = epe_stats(u, v, u0, v0)
        out[name]["seconds"] = round(time.time() - t, 1)
    return out


def main():
    from oracle import oracle as O
    from conftest import read_ppm
    from eppm_amd import synth
    G = os.path.join(ROOT, "tests", "golden")
    f10, f11 = read_ppm(os.path.join(G, "frame10.ppm")), read_ppm(os.path.join(G, "frame11.ppm"))
    if "--small" in sys.argv:
        print(json.dumps(envelope(f10[180:300, 240:400].copy(), f11[180:300, 240:400].copy(), O), indent=1))
        return
    res = {"reference": "lockstep oracle (oracle/eppm_oracle.c, all variants off)", "unit": "px, end-point error of the variant's flow against the lockstep oracle's",
           "cases": {}}
    res["cases"]["middlebury_640x480"] = envelope(f10, f11, O)
    a, b, gu, gv = synth.make_pair(436, 1024, seed=1234)
    res["cases"]["sintel_shape_1024x436_seed1234"] = envelope(a, b, O)
    # for scale: the lockstep oracle's own error against the synthetic ground truth
    u0, v0 = O.compute_flow(a, b)
    res["cases"]["sintel_shape_1024x436_seed1234"]["lockstep_vs_ground_truth"] = epe_stats(u0, v0, gu, gv)
    json.dump(res, open(os.path.join(ROOT, "profiles", "parity_envelope.json"), "w"), indent=1)
    for case, r in res["cases"].items():
        print(case)
        for name, st in r.items():
            print(f"  {name:26s} mean {st['mean_epe_px']:.4f}  p95 {st['p95_epe_px']:.3f}  max {st['max_epe_px']:.1f}  differing {100 * st['frac_differing']:.1f} %  >1px {100 * st['frac_over_1px']:.2f} %")


if __name__ == "__main__":
    main()
