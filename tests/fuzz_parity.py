"""Randomized parity sweep (run by hand on a GPU box: python tests/fuzz_parity.py SEED N): random sizes, parameters and
image statistics, HIP vs oracle, bit-exact.  Lives under tests/ because it uses the oracle (test infrastructure)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eppm_amd
from eppm_amd import synth
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for t in range(n):
    h, w = int(rng.integers(16, 200)), int(rng.integers(16, 260))
    kind = t % 4
    if kind == 0:
        a, b, _, _ = synth.make_pair(h, w, seed=int(rng.integers(1 << 30)), max_flow=float(rng.uniform(1, 30)))
    elif kind == 1:   # pure noise, unrelated images
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8); b = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 2:   # flat regions + saturated blocks
        a = np.zeros((h, w, 3), np.uint8); a[h // 3:, w // 4:] = 255; a[: h // 2, : w // 2, 1] = 128
        b = np.roll(a, (int(rng.integers(-9, 9)), int(rng.integers(-9, 9))), axis=(0, 1))
    else:             # low contrast
        base = rng.integers(100, 110, (h, w, 3)).astype(np.uint8)
        a = base; b = np.roll(base, 3, axis=1)
    params = dict(patch_r=int(rng.choice([9, 9, 9, 17, 5, 4])), num_iter=int(rng.integers(1, 5)), num_guess=int(rng.integers(1, 9)),
                  seg_len=int(rng.integers(2, 14)), wmf_iters=int(rng.integers(0, 6)), search_range=int(rng.integers(1, 40)),
                  seed=int(rng.integers(1, 1 << 40)), propagation=int(rng.integers(0, 3)), levels=int(rng.integers(1, 5)))
    e = eppm_amd.EPPM(params=eppm_amd.Params(**params))
    e.init(a, b, h, w)
    u, v = e.compute_flow()
    ou, ov = O.compute_flow(a, b, O.default_params(**params))
    same = np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(v.view(np.uint32), ov.view(np.uint32))
    nan = int(np.isnan(u).sum())
    print(t, (h, w), params, "OK" if same else "MISMATCH", "nan", nan, flush=True)
    bad += (not same)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
