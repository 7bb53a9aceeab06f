"""Who causes the evaluations a late merged phase A still has to do, and how long the merged form's lists are (CPU oracle; test
infrastructure).

Phase A of iteration k + 1 tests, for every pixel i and direction d, the candidate shift_d(F[i - 1_d]) of the field F the iteration starts
with.  It is answered without an evaluation when it equals the pixel's own match (skip rule) or the candidate this pixel was tested with
in iteration k (evaluation cache) -- so it needs an evaluation only where the neighbour's match CHANGED during iteration k: by one of
iteration k's sweeps (known before search k starts) or by search k itself (known only after it).  A phase A started under search k on the
field before that search can therefore pre-evaluate the first kind only.  Printed per iteration: evaluations per 1000 pixel-directions,
the share caused by the sweeps, by the search; and the units a direction lists (pixels whose candidate would be accepted -> unit = two
adjacent segments of the line), per problem, for sizing the one-launch form of the four in-place sweeps.
usage: phase_a_sources.py [WxH [seed [case]]]   case: synth (default) | noise | same | shift40"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from eppm_amd import synth  # noqa: E402

O.set_num_threads(min(8, os.cpu_count() or 1))
w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024x436").split("x"))
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
case = sys.argv[3] if len(sys.argv) > 3 else "synth"
a, b, _, _ = synth.make_pair(h, w, seed=seed)
rng = np.random.default_rng(seed)
if case == "noise":
    a = rng.integers(0, 256, a.shape, dtype=np.uint8)
    b = rng.integers(0, 256, a.shape, dtype=np.uint8)
elif case == "same":
    b = a.copy()
elif case == "shift40":
    b = np.roll(a, (12, 40), (0, 1))
_, _, st = O.compute_flow(a, b, dump=True)
i1, i2, c1, c2 = st["img1_L2"], st["img2_L2"], st["cen1_L2"], st["cen2_L2"]
H, W = i1.shape
SL = 10


def cands(F):
    """candidate (x, y) of every pixel for the four directions from field F; invalid (-1) where the pixel has no predecessor"""
    out = []
    for d in range(4):
        cx = np.full((H, W), -1, np.int32); cy = np.full((H, W), -1, np.int32)
        fx, fy = F["x"].astype(np.int32), F["y"].astype(np.int32)
        if d == 0: cx[:, 1:] = np.minimum(fx[:, :-1] + 1, W - 1); cy[:, 1:] = fy[:, :-1]
        elif d == 1: cx[1:, :] = fx[:-1, :]; cy[1:, :] = np.minimum(fy[:-1, :] + 1, H - 1)
        elif d == 2: cx[:, :-1] = np.maximum(fx[:, 1:] - 1, 0); cy[:, :-1] = fy[:, 1:]
        else: cx[:-1, :] = fx[1:, :]; cy[:-1, :] = np.maximum(fy[1:, :] - 1, 0)
        out.append((cx, cy))
    return out


def shift_back(M, d):
    """M at the predecessor of every pixel in direction d"""
    R = np.zeros_like(M)
    if d == 0: R[:, 1:] = M[:, :-1]
    elif d == 1: R[1:, :] = M[:-1, :]
    elif d == 2: R[:, :-1] = M[:, 1:]
    else: R[:-1, :] = M[1:, :]
    return R


nnf, states = O.gen_rand_field(W, H)
cost = O.cost_field(nnf, i1, i2, c1, c2)
prev_c = None
print(f"{case} {w}x{h} seed {seed}: level-2 field {W}x{H}")
for it in range(10):
    F0 = nnf.copy()
    c0 = cands(F0)
    listed = []
    for d in range(4):
        ncost, nn = O.seg_propagate_dir(cost, nnf, i1, i2, c1, c2, d)
        acc = (nn["x"] != nnf["x"]) | (nn["y"] != nnf["y"])
        along = np.arange(W)[None, :] if d in (0, 2) else np.arange(H)[:, None]
        line = np.arange(H)[:, None] if d in (0, 2) else np.arange(W)[None, :]
        seg = np.where((d < 2) & (along < SL), 0, along // SL)
        nseg = ((W if d in (0, 2) else H) + SL - 1) // SL
        unit = line * ((nseg + 1) // 2) + seg // 2 + 0 * along
        listed.append(len(np.unique(np.broadcast_to(unit, acc.shape)[acc])))      # lower bound: units with an ACCEPTED pixel
        cost, nnf = ncost, nn
    Fs = nnf.copy()
    states, cost, nnf = O.random_search(states, cost, nnf, i1, i2, c1, c2)
    F1 = nnf
    c1n = cands(F1)
    if it >= 3:
        tot = ev = by_sweep = by_search = 0
        for d in range(4):
            valid = c1n[d][0] >= 0
            own = (c1n[d][0] == F1["x"]) & (c1n[d][1] == F1["y"])
            cached = (c1n[d][0] == c0[d][0]) & (c1n[d][1] == c0[d][1])
            need = valid & ~own & ~cached
            sw = shift_back((Fs["x"] != F0["x"]) | (Fs["y"] != F0["y"]), d)
            se = shift_back((F1["x"] != Fs["x"]) | (F1["y"] != Fs["y"]), d)
            tot += valid.sum(); ev += need.sum(); by_search += (need & se).sum(); by_sweep += (need & sw & ~se).sum()
        print(f"it {it}: phase A of it {it + 1}: {1000 * ev / tot:6.1f} evaluations per 1000 pixel-directions; caused by sweeps of it {it}: {100 * by_sweep / max(ev, 1):4.1f} %, "
              f"by search {it}: {100 * by_search / max(ev, 1):4.1f} %;  units with an accepted pixel in it {it}: {listed} of {H * ((W + SL - 1) // SL + 1) // 2} / {W * ((H + SL - 1) // SL + 1) // 2}")
