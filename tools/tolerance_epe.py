"""End-point error of the tolerance library libeppm_hip_tol.so (integer-domain tables instead of the two software exp of the patch
term, fused sums; NOT bit-identical) against the exact library -- which equals the CPU oracle bit for bit (tests/), so
this is the EPE against the oracle at sizes the oracle cannot be re-run at on the GPU box.  Cases: the bundled frame10/frame11
pair forwards (north_star's tolerance case: <= 1e-3 px mean EPE) and backwards, BASELINE configs[1] (1024x436), configs[3]
(1920x1080), configs[4] (3840x2160, radius 17; only with --all) and the eight fixed-seed fuzz cases of tests/test_configs_gpu.py.
Prints one JSON line.

usage: tolerance_epe.py [--all] [--no-fuzz] [--fuzz-seeds FIRST:N]        (internal: --dump FILE computes the flows with the library EPPM_HIP_VARIANT selects)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def cases(all_sizes):
    from conftest import read_ppm, GOLDEN
    from eppm_amd import synth
    import test_configs_gpu as T
    a, b = read_ppm(os.path.join(GOLDEN, "frame10.ppm")), read_ppm(os.path.join(GOLDEN, "frame11.ppm"))
    out = [("bundled_640x480", a, b, {}), ("bundled_640x480_backwards", b, a, {}),
           ("config2_1024x436", *synth.make_pair(436, 1024, seed=1234)[:2], {}),
           ("config4_1920x1080", *synth.make_pair(1080, 1920, seed=1234, max_flow=40.0)[:2], {})]
    if all_sizes:
        out.append(("config5_3840x2160_r17", *synth.make_pair(2160, 3840, seed=1234, max_flow=60.0)[:2], dict(patch_r=17)))
    if "--no-fuzz" not in sys.argv:
        # the fixed-seed fuzz cases of the parity suite, all eight kinds per seed: synthetic motion, unrelated noise, flat regions with
        # saturated blocks (costs that tie exactly, weights that underflow), low contrast; odd sizes, radii 4 / 5 / 9 / 17, 1-4 iterations,
        # 1-8 guesses, every propagation mode and pyramid depth
        seeds = T.FUZZ_SEEDS
        if "--fuzz-seeds" in sys.argv:          # FIRST:N -- a wider sweep than the suite's eight seeds (outside the suite)
            first, n = (int(x) for x in sys.argv[sys.argv.index("--fuzz-seeds") + 1].split(":"))
            seeds = range(first, first + n)
        for seed in seeds:
            for t in range(8):
                fa, fb, params = T._fuzz_case(seed, t)
                out.append((f"fuzz_seed{seed}_case{t}", fa, fb, params))
    return out


def flows(all_sizes):
    import eppm_amd
    res = {}
    todo = cases(all_sizes)
    # (importing tests/conftest.py selected the parity tests' library for this process: select the one this run is about)
    eppm_amd.select_library(os.environ.get("EPPM_HIP_VARIANT", "") or "")
    for name, a, b, params in todo:
        h, w, _ = a.shape
        e = eppm_amd.EPPM(params=eppm_amd.Params(**params) if params else None)
        e.init(a, b, h, w)
        u, v = e.compute_flow()
        e.close()
        res[name + "_u"], res[name + "_v"] = u, v
    return res, eppm_amd.lib().eppm_version().decode()


def main():
    all_sizes = "--all" in sys.argv
    if "--dump" in sys.argv:
        res, ver = flows(all_sizes)
        np.savez(sys.argv[sys.argv.index("--dump") + 1], version=np.array(ver), **res)
        return
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "tol.npz")
        env = dict(os.environ, EPPM_HIP_VARIANT="tol")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", f] + [x for x in ("--all", "--no-fuzz") if x in sys.argv] + (sys.argv[sys.argv.index("--fuzz-seeds"):][:2] if "--fuzz-seeds" in sys.argv else []), env=env, check=True)
        ap = np.load(f)
        approx = {k: ap[k] for k in ap.files}
    os.environ.pop("EPPM_HIP_VARIANT", None)
    exact, ver = flows(all_sizes)
    out = {"library": str(approx.pop("version")), "against": ver + " (= the CPU oracle, bit for bit)", "tolerance_px": 1e-3, "cases": {}}
    for k in sorted(k[:-2] for k in exact if k.endswith("_u")):
        epe = np.sqrt((approx[k + "_u"].astype(np.float64) - exact[k + "_u"]) ** 2 + (approx[k + "_v"].astype(np.float64) - exact[k + "_v"]) ** 2)
        out["cases"][k] = {"epe_mean_px": float(epe.mean()), "epe_p99_px": float(np.percentile(epe, 99)), "epe_max_px": float(epe.max()),
                           "pixels_differing": float((epe > 0).mean()), "frac_over_1px": float((epe > 1.0).mean()), "pixels": int(epe.size)}
    # the keys bench.py and the test read
    out["pair"] = "frame10/11 640x480"
    out["epe_mean_px"] = out["cases"]["bundled_640x480"]["epe_mean_px"]
    out["epe_max_px"] = out["cases"]["bundled_640x480"]["epe_max_px"]
    out["worst_case_mean_px"] = max(c["epe_mean_px"] for c in out["cases"].values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
