"""Extended fixed-seed fuzz of the whole path against the live CPU oracle (GPU box; not part of the suite: ~10 minutes): the cases of
tests/test_configs_gpu.py::_fuzz_case for seeds first .. first+N-1 (default 100), each through a single-pair context with the sweeps' form left to the
library (classic + evaluation cache at these sizes) and forced speculative (two-launch form with and without the work list, merged form), and through a 3-pair batch context (pair,
reversed pair, pair) forced speculative in both forms.  Prints one line per failure and a summary; exit code 1 on any mismatch.
usage: fuzz_extended.py [N [first_seed]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np  # noqa: E402
import eppm_amd  # noqa: E402
import test_configs_gpu as T  # noqa: E402
from oracle import oracle as O  # noqa: E402


def same(got, want):
    return np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    eppm_amd.select_library("test")          # the sweeps' forms are forced through eppm_test_set_option (include/eppm_test.h)
    L = eppm_amd.lib()
    bad = runs = 0
    for seed in range(first, first + n):
        for t in range(8):
            a, b, params = T._fuzz_case(seed, t)
            h, w, _ = a.shape
            want = O.compute_flow(a, b, O.default_params(**params))
            rev = O.compute_flow(b, a, O.default_params(**params))
            for mode in (-1, 1, 2, 3):       # library default, speculative with the work list, without it, merged form from the first iteration
                L.eppm_test_set_option(b"sweep_spec", mode)
                try:
                    e = eppm_amd.EPPM(params=eppm_amd.Params(**params))
                    e.init(a, b, h, w)
                    got = e.compute_flow()
                    e.set_data(b, a)                      # a second run through the same context: the evaluation cache must start empty
                    got_rev = e.compute_flow()
                    e.close()
                finally:
                    L.eppm_test_set_option(b"sweep_spec", -1)
                runs += 2
                if not same(got, want) or not same(got_rev, rev):
                    bad += 1
                    print(f"MISMATCH seed {seed} case {t} sweep_spec {mode}: {w}x{h} {params}", flush=True)
            for mode in (1, 3):
                L.eppm_test_set_option(b"sweep_spec", mode)
                try:
                    B = eppm_amd.EPPMBatch(h, w, 3, params=eppm_amd.Params(**params))
                    B.set_data([(a, b), (b, a), (a, b)])
                    out = B.compute_flow()
                    B.close()
                finally:
                    L.eppm_test_set_option(b"sweep_spec", -1)
                runs += 3
                if not (same(out[0], want) and same(out[1], rev) and same(out[2], want)):
                    bad += 1
                    print(f"MISMATCH seed {seed} case {t} batch sweep_spec {mode}: {w}x{h} {params}", flush=True)
    print(f"fuzz_extended: {runs} flows checked against the oracle, {bad} mismatching cases")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
