#!/usr/bin/env python3
"""How far does the tolerance arithmetic of libeppm_hip_tol.so move the flow?  (CPU only; test infrastructure.)

north_star allows floating-point work "within 1e-3 px EPE on the bundled Middlebury pair".  The tolerance library replaces the two
software exp of the patch term by integer-domain tables / one hardware exp2 and fuses its sums; this script runs the CPU oracle with the
same substitutions (oracle/eppm_oracle.c: orc_set_tol_variant) and reports the end-point error of each against the lockstep
oracle, per stage scope, so that a step which would leave the tolerance is known before a kernel is written.

usage: tolerance_envelope.py [--small] [--synthetic]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (name, mode, scope)
VARIANTS = [("tables_pm", 1, 1), ("tables_refine", 1, 2), ("tables_smoothing", 1, 4),
            ("tables_fma_pm_refine", 3, 3), ("kernel_form_tables_fma_chunks_exp2_refine", 23, 3), ("tables_fma_all", 3, 7)]


def epe_stats(u, v, u0, v0):
    e = np.sqrt((u.astype(np.float64) - u0) ** 2 + (v.astype(np.float64) - v0) ** 2)
    return {"mean_epe_px": float(e.mean()), "p99_epe_px": float(np.percentile(e, 99)), "max_epe_px": float(e.max()),
            "frac_differing": float((e > 0).mean()), "frac_over_1px": float((e > 1.0).mean())}


def envelope(a, b, O, variants=VARIANTS):
    O.set_tol_variant()
    u0, v0 = O.compute_flow(a, b)
    out = {}
    for name, mode, scope in variants:
        O.set_tol_variant(mode, scope)
        t = time.time()
        try:
            u, v = O.compute_flow(a, b)
        finally:
            O.set_tol_variant()
        out[name] = epe_stats(u, v, u0, v0)
        out[name]["seconds"] = round(time.time() - t, 1)
        print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
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
    res = {"reference": "lockstep oracle (all variants off)", "unit": "px, end-point error against the lockstep oracle", "cases": {}}
    res["cases"]["middlebury_640x480"] = envelope(f10, f11, O)
    res["cases"]["middlebury_640x480_backwards"] = envelope(f11, f10, O)
    if "--synthetic" in sys.argv:
        a, b, _, _ = synth.make_pair(436, 1024, seed=1234)
        res["cases"]["sintel_shape_1024x436_seed1234"] = envelope(a, b, O)
    json.dump(res, open(os.path.join(ROOT, "profiles", "tolerance_envelope.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
