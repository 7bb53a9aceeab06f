"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same inputs.
Bar: bit-exact for every plane (u8 / int16 / float32 alike) -- the float formulas are shared by
construction, so exact equality is the test; the end-to-end EPE bound of
BASELINE.json's north_star (1e-3 px) follows trivially and is asserted as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def eq(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: shape/dtype {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    if a.dtype.kind == "f":
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    else:
        same = a.view(np.uint8) == b.view(np.uint8)
    n = int((~same).sum())
    assert n == 0, f"{what}: {n} of {same.size} elements differ"


@pytest.fixture(scope="module")
def S():
    import eppm_amd
    eppm_amd.lib()
    from eppm_amd import stages
    stages.set_params(None)
    return stages


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def test_fast_exp_bits(S, O):
    # dense over the gradual-underflow range (results below 2^-126 from x = -87.3, zero from x = -104.3) and far beyond
    x = -np.concatenate([np.linspace(0, 120, 20001), np.linspace(86, 106, 40001), np.linspace(120, 4000, 2001),
                         np.float32(np.arange(0, 256)) ** 2 / np.float32(255 * 255 * 0.01)]).astype(np.float32)
    eq(S.probe_fast_exp(x), O.fast_exp(x), "fast_exp")


def test_div_const_bits(S):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(200000, dtype=np.float32) * 2, [0.0]]).astype(np.float32)
    eq(S.probe_div_const(x, 0), (x / (np.float32(0.1) * np.float32(0.1))).astype(np.float32), "x/(.1f*.1f)")
    eq(S.probe_div_const(x, 1), (x / (np.float32(0.02) * np.float32(0.02))).astype(np.float32), "x/(.02f*.02f)")
    c = np.arange(256, dtype=np.float32)
    eq(S.probe_div_const(c, 2), (c / np.float32(255)).astype(np.float32), "unorm8")


def test_range_terms_by_table_equal_the_formula_for_every_byte_pair(S, O):
    """The patch data term 1 - exp(-d^2 / LAMBDA_AD^2) and the smoothing / weighted-median weight exp(-d^2 / SIG_R^2) are READ from a table
    of the possible L-inf distances of unorm8 texels (eppm_device.cuh: DeltaTab; 598 distinct floats over the 65 536 byte pairs) instead of
    being evaluated.  For every byte pair the table must return the bits the oracle's formula returns."""
    g = (np.arange(256, dtype=np.float32) / np.float32(255)).astype(np.float32)
    a, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    d = np.abs(g[a] - g[b]).astype(np.float32).reshape(-1)
    assert len(np.unique(d)) == 598
    uniq, inv = np.unique(d, return_inverse=True)
    for which, s in ((0, np.float32(0.1) * np.float32(0.1)), (1, np.float32(0.02) * np.float32(0.02))):
        arg = (-(uniq * uniq) / s).astype(np.float32)                      # IEEE float32, as the oracle forms it
        e = O.fast_exp(arg)
        want = ((np.float32(1) - e) if which == 0 else e).astype(np.float32)[inv]
        eq(S.probe_delta_table(d, which), want, f"range term {which} by table")


def test_prepare_stages(S, O, crop):
    raw = O.rgb2rgba(crop[0])
    eq(S.gauss_filter_rgba(raw, 0.5, 2), O.gauss_filter_rgba(raw, 0.5, 2), "gauss s=.5 r=2")
    eq(S.gauss_filter_rgba(raw, 2.0, 6), O.gauss_filter_rgba(raw, 2.0, 6), "gauss s=2 r=6")
    eq(S.resize_rgba(raw, 60, 80, 0.5), O.resize_rgba(raw, 60, 80, 0.5), "resize 1/2")
    eq(S.resize_rgba(raw, 30, 40, 0.25), O.resize_rgba(raw, 30, 40, 0.25), "resize 1/4")
    c1, c2 = S.census_transform(raw, O.rgb2rgba(crop[1]))
    eq(c1, O.census(raw), "census 1")
    eq(c2, O.census(O.rgb2rgba(crop[1])), "census 2")


def test_prepare_launcher(S, O, crop, crop_stages):
    st = crop_stages
    dims = list(zip(st["arrH"], st["arrW"]))
    i1, i2, c1, c2 = S.prepare(O.rgb2rgba(crop[0]), O.rgb2rgba(crop[1]), dims)
    for l in range(3):
        eq(i1[l], st[f"img1_L{l}"], f"img1 L{l}")
        eq(i2[l], st[f"img2_L{l}"], f"img2 L{l}")
        eq(c1[l], st[f"cen1_L{l}"], f"census1 L{l}")
        eq(c2[l], st[f"cen2_L{l}"], f"census2 L{l}")


@pytest.fixture(scope="module")
def L1(crop_stages):
    """Level-1 planes of the crop (80x60): small enough for per-kernel PatchMatch parity."""
    st = crop_stages
    return st["img1_L1"], st["img2_L1"], st["cen1_L1"], st["cen2_L1"]


def test_patchmatch_substages(S, O, L1):
    i1, i2, c1, c2 = L1
    h, w = i1.shape
    P = S.PlaneSet(i1, i2, c1, c2)
    rng = S.PmRng(w, h)
    nnf = S.pm_gen_rand_field(rng)
    onnf, ostates = O.gen_rand_field(w, h)
    eq(nnf, onnf, "random NNF")
    eq(rng.block_states(), ostates, "RNG states after init field")
    cost = S.pm_cost_field(nnf, P)
    ocost = O.cost_field(onnf, i1, i2, c1, c2)
    eq(cost, ocost, "initial cost field")
    import eppm_amd
    L = eppm_amd.lib()
    for it in range(3):
        for d in range(4):
            # both forms of the sweep on the same state: classic, and speculative (phase A + phase B; the context path uses it
            # from the third iteration on (EPPM_SPEC_FROM_ITER = 2), here it also meets the first iterations' many accepted candidates)
            spec = {}
            for mode in (1, 2):         # 1: phase B walks the chains phase A listed (work list); 2: phase B walks every chain
                try:
                    assert L.eppm_test_set_option(b"sweep_spec", mode) == 0
                    spec[mode] = S.pm_seg_propagate(cost, nnf, P, d)
                finally:
                    L.eppm_test_set_option(b"sweep_spec", -1)
            scost, snnf = spec[1]
            cost, nnf = S.pm_seg_propagate(cost, nnf, P, d)
            ocost, onnf = O.seg_propagate_dir(ocost, onnf, i1, i2, c1, c2, d)
            eq(nnf, onnf, f"NNF after propagate dir {d} iter {it}")
            eq(cost, ocost, f"cost after propagate dir {d} iter {it}")
            eq(snnf, onnf, f"NNF after speculative propagate dir {d} iter {it}")
            eq(scost, ocost, f"cost after speculative propagate dir {d} iter {it}")
            eq(spec[2][1], onnf, f"NNF after speculative propagate without work list dir {d} iter {it}")
            eq(spec[2][0], ocost, f"cost after speculative propagate without work list dir {d} iter {it}")
        cost, nnf = S.pm_random_search(rng, cost, nnf, P)
        ostates, ocost, onnf = O.random_search(ostates, ocost, onnf, i1, i2, c1, c2)
        eq(nnf, onnf, f"NNF after random search iter {it}")
        eq(cost, ocost, f"cost after random search iter {it}")
        eq(rng.block_states(), ostates, f"RNG states after search iter {it}")


def test_patchmatch_evaluation_kernels_with_arbitrary_nnf(S, O, L1):
    """An NNF as a caller of the stage launchers may hand over -- targets on the last row / column, one past them (the
    reference's inclusive random range), and far outside the image: the cost field and the speculative sweeps (phase A's
    one-evaluation-per-lane kernel, phase B's walk) clamp every sample as the texture model does."""
    import eppm_amd
    i1, i2, c1, c2 = L1
    h, w = i1.shape
    P = S.PlaneSet(i1, i2, c1, c2)
    rng = np.random.default_rng(77)
    nnf = np.zeros((h, w), O.short2)
    nnf["x"] = rng.integers(0, w + 1, (h, w))
    nnf["y"] = rng.integers(0, h + 1, (h, w))
    m = rng.random((h, w))
    nnf["x"][m < 0.05] = w; nnf["y"][(m > 0.05) & (m < 0.1)] = h               # one past the last column / row
    nnf["x"][(m > 0.1) & (m < 0.13)] = -7; nnf["y"][(m > 0.13) & (m < 0.16)] = h + 40      # outside: gather path
    nnf["x"][(m > 0.16) & (m < 0.18)] = w + 300
    cost = O.cost_field(nnf, i1, i2, c1, c2)
    eq(S.pm_cost_field(nnf, P), cost, "cost field of the arbitrary NNF")
    L = eppm_amd.lib()
    ocost, onnf = cost, nnf
    try:
        assert L.eppm_test_set_option(b"sweep_spec", 1) == 0
        for d in range(4):
            cost, nnf = S.pm_seg_propagate(cost, nnf, P, d)
            ocost, onnf = O.seg_propagate_dir(ocost, onnf, i1, i2, c1, c2, d)
            eq(nnf, onnf, f"NNF after speculative propagate dir {d}")
            eq(cost, ocost, f"cost after speculative propagate dir {d}")
    finally:
        L.eppm_test_set_option(b"sweep_spec", -1)


def test_patchmatch_launcher(S, O, crop_stages):
    st = crop_stages
    P = S.PlaneSet(st["img1_L2"], st["img2_L2"], st["cen1_L2"], st["cen2_L2"])
    nnf, cost = S.patchmatch(P)
    eq(nnf, st["nnf1_pm"], "baoCudaPatchMatch NNF (10 iterations)")
    eq(cost, st["cost1_pm"], "baoCudaPatchMatch cost")
    P2 = S.PlaneSet(st["img2_L2"], st["img1_L2"], st["cen2_L2"], st["cen1_L2"])
    nnf2, cost2 = S.patchmatch(P2)
    eq(nnf2, st["nnf2_pm"], "backward NNF")
    eq(cost2, st["cost2_pm"], "backward cost")


def test_level2_post(S, O, crop_stages):
    st = crop_stages
    a, b, c, d = S.left_right_check(st["nnf1_pm"], st["cost1_pm"], st["nnf2_pm"], st["cost2_pm"])
    oa, ob, oc, od = O.left_right_check(st["nnf1_pm"], st["cost1_pm"], st["nnf2_pm"], st["cost2_pm"])
    eq(a, oa, "LR nnf1"); eq(b, ob, "LR cost1"); eq(c, oc, "LR nnf2"); eq(d, od, "LR cost2")
    eq(a, st["nnf1_lr"], "LR vs pipeline dump")
    n2, c2 = S.outlier_removal(a, b)
    on2, oc2 = O.outlier_removal(oa, ob)
    eq(n2, on2, "outlier nnf"); eq(c2, oc2, "outlier cost")
    img = st["img1_L2"]
    for iters in (1, 3, 20):
        eq(S.weighted_median(n2, img, iters, True), O.weighted_median(on2, img, iters, True), f"WMF x{iters} occlusion only")
    eq(S.weighted_median(n2, img, 1, False), O.weighted_median(on2, img, 1, False), "WMF all pixels")
    eq(S.fill_holes(n2, img), O.fill_holes(on2, img), "fill holes (before WMF: many holes)")
    eq(S.fill_holes(st["nnf1_wmf"], img), st["nnf1_fill"], "fill holes")
    eq(S.nnf2flow(st["nnf1_fill"]), st["flow_L2"], "NNF -> flow")
    eq(S.nnf2flow(n2), O.nnf2flow(on2), "NNF -> flow with invalid pixels")


def test_weighted_median_long_lists(S, O):
    """A front that fills 4 pixels per launch from one valid column: the work list stays tens of thousands of entries
    long for every launch (several append batches per workgroup once the grids shrink) and every launch fills pixels,
    then the field reaches its fixed point and the remaining launches exit early."""
    rng = np.random.default_rng(3)
    h, w = 200, 256
    img = O.rgb2rgba(rng.integers(90, 110, (h, w, 3), dtype=np.uint8))          # low contrast: non-zero bilateral weights
    nnf = np.zeros((h, w), O.short2)
    nnf["x"][:] = -10000; nnf["y"][:] = -10000
    ys, xs = np.mgrid[0:h, 0:w]
    nnf["x"][:, :2] = (xs[:, :2] + rng.integers(0, 4, (h, 2))).astype(np.int16)
    nnf["y"][:, :2] = (ys[:, :2] + rng.integers(0, 4, (h, 2))).astype(np.int16)
    for iters in (1, 7, 20):
        got, want = S.weighted_median(nnf, img, iters, True), O.weighted_median(nnf, img, iters, True)
        eq(got, want, "front after %d launches" % iters)
    assert (want["x"] >= 0).sum() > (nnf["x"] >= 0).sum() * 10
    # fixed point long before the last launch: a field that nothing can fill
    dead = nnf.copy(); dead["x"][:] = -10000; dead["y"][:] = -10000
    eq(S.weighted_median(dead, img, 20, True), dead, "nothing to fill")


def test_c2f(S, O, crop_stages):
    st = crop_stages
    up = S.resize_flow(st["flow_L2"], st["arrH"][1], st["arrW"][1], 2.0)
    eq(up, O.resize_flow(st["flow_L2"], st["arrH"][1], st["arrW"][1], 2.0), "flow upsample x2")
    P1 = S.PlaneSet(st["img1_L1"], st["img2_L1"], st["cen1_L1"], st["cen2_L1"])
    f1 = S.blf_c2f(st["flow_L2"], P1, (st["arrH"][2], st["arrW"][2]))
    eq(f1, st["flow_c2f_L1"], "baoCudaBLF_C2F level 1")
    s1 = S.flow_smoothing(f1, st["img1_L1"])
    eq(s1, st["flow_L1"], "flow smoothing level 1")
    # unknown flow handling: plant 1e10 vectors
    fl = st["flow_L1"].copy()
    fl["x"][5:9, 7:30] = 1e10
    fl["y"][5:9, 7:30] = 1e10
    eq(S.flow_smoothing(fl, st["img1_L1"]), O.flow_smoothing(fl, st["img1_L1"]), "smoothing with unknown flow")
    eq(S.c2f_refine(fl, P1), O.c2f_refine(fl, st["img1_L1"], st["img2_L1"], st["cen1_L1"], st["cen2_L1"]), "refine with unknown flow")


def test_c2f_refine_window_and_fallback_paths(S, O, crop_stages):
    """k_c2f_refine_win stages the target window of a tile in LDS when the tile's candidate centres are coherent and falls back
    to per-access gathers otherwise.  Flows that exercise both inside one launch: constant, tiles at the exact admissible
    spread (33 x 25) and one past it, per-pixel random jumps, vectors pointing far outside the image (window rows and columns
    clamped at load), and unknown vectors mixed in."""
    st = crop_stages
    i1, i2, c1, c2 = st["img1_L0"], st["img2_L0"], st["cen1_L0"], st["cen2_L0"]
    P0 = S.PlaneSet(i1, i2, c1, c2)
    h, w = i1.shape
    rng = np.random.default_rng(21)

    def run(fx, fy, what):
        f = np.zeros((h, w), O.float2)
        f["x"], f["y"] = fx.astype(np.float32), fy.astype(np.float32)
        eq(S.c2f_refine(f, P0), O.c2f_refine(f, i1, i2, c1, c2), what)

    z = np.zeros((h, w))
    run(z + 3.7, z - 2.2, "constant flow (coherent everywhere)")
    # The admissible spread of a tile's candidate centres (eppm_probe_c2f_window).  Shifting the LAST column (row) of every
    # 16x16 tile by d makes the spread exactly 15 + d: one below the limit, at the limit (the window's last column / row is
    # read), and one past it (fallback).  An off-by-one here reads a texel of the next window row: caught bit for bit.
    import ctypes as C
    from eppm_amd import lib
    sx, sy = C.c_int(), C.c_int()
    assert lib().eppm_probe_c2f_window(9, C.byref(sx), C.byref(sy)) == 0
    for dxs, dys in ((-1, -1), (0, 0), (1, 1), (0, -40), (-40, 0), (1, -40), (-40, 1)):
        fx, fy = z.copy(), z.copy()
        fx[:, 15::16] = sx.value - 15 + dxs
        fy[15::16, :] = sy.value - 15 + dys
        run(fx, fy, "centre spread = limit %+d (x), limit %+d (y)" % (dxs, dys))
        fx, fy = z.copy(), z.copy()                   # the same with the first column / row pulled the other way
        fx[:, 0::16] = -(sx.value - 15 + dxs)
        fy[0::16, :] = -(sy.value - 15 + dys)
        run(fx, fy, "centre spread = limit %+d (x), limit %+d (y), negative side" % (dxs, dys))
    run(rng.integers(-40, 41, (h, w)), rng.integers(-40, 41, (h, w)), "random jumps (incoherent)")
    run(z - 300.0, z + 250.0, "targets far outside the image")
    fx, fy = rng.normal(0, 1.5, (h, w)) + 5, rng.normal(0, 1.5, (h, w)) - 4
    m = rng.random((h, w)) < 0.1
    fx[m] = 1e10
    fy[m] = 1e10
    fx[:16, :16] = 1e10            # a tile without any known pixel
    fy[:16, :16] = 1e10
    run(fx, fy, "noisy flow with unknown vectors")
    # ragged sizes (no dimension a multiple of 16, images smaller than a window): tiles cut by the image edge, windows
    # clamped on every side
    for (hh, ww) in ((77, 101), (33, 250), (130, 47), (17, 19)):
        a = np.zeros((hh, ww), O.uchar4)
        b = np.zeros((hh, ww), O.uchar4)
        for ch in ("x", "y", "z"):
            base = rng.integers(0, 256, (hh, ww))
            a[ch] = base
            b[ch] = np.roll(base, (2, -3), axis=(0, 1))
        ca, cb = O.census(a), O.census(b)
        Pr = S.PlaneSet(a, b, ca, cb)
        f = np.zeros((hh, ww), O.float2)
        f["x"] = (rng.normal(0, 2.0, (hh, ww)) - 3).astype(np.float32)
        f["y"] = (rng.normal(0, 2.0, (hh, ww)) + 2).astype(np.float32)
        eq(S.c2f_refine(f, Pr), O.c2f_refine(f, a, b, ca, cb), "ragged %dx%d" % (ww, hh))
    # NaN costs (black source, white target: every range weight underflows to 0, cost = 0/0): the nested __min and the strict <
    # of the candidate loop must treat them as the reference's expressions do
    a = np.zeros((64, 96), O.uchar4)
    b = np.zeros((64, 96), O.uchar4)
    for ch in ("x", "y", "z"):
        b[ch] = 255
    b["x"][20:40, 30:60] = 0          # a patch where some candidates do get finite costs
    b["y"][20:40, 30:60] = 0
    b["z"][20:40, 30:60] = 0
    ca, cb = O.census(a), O.census(b)
    f = np.zeros((64, 96), O.float2)
    f["x"] = rng.integers(-2, 3, (64, 96)).astype(np.float32)
    f["y"] = rng.integers(-2, 3, (64, 96)).astype(np.float32)
    eq(S.c2f_refine(f, S.PlaneSet(a, b, ca, cb)), O.c2f_refine(f, a, b, ca, cb), "NaN costs")
    # patch radius 17: k_c2f_refine_win4 (1024-thread workgroups, four pass groups, 80x72-texel window)
    import eppm_amd
    p17 = eppm_amd.Params(patch_r=17)
    S.set_params(p17)
    try:
        assert lib().eppm_probe_c2f_window(17, C.byref(sx), C.byref(sy)) == 0
        for dxs, dys in ((-1, -1), (0, 0), (1, 1), (0, -40), (-40, 0)):
            fx, fy = z.copy(), z.copy()
            fx[:, 15::16] = sx.value - 15 + dxs
            fy[15::16, :] = sy.value - 15 + dys
            f = np.zeros((h, w), O.float2)
            f["x"], f["y"] = fx.astype(np.float32), fy.astype(np.float32)
            eq(S.c2f_refine(f, P0), O.c2f_refine(f, i1, i2, c1, c2, O.default_params(patch_r=17)), "R=17 centre spread = limit %+d (x), %+d (y)" % (dxs, dys))
        for fx, fy, what in ((z + 2.0, z - 1.0, "R=17 constant flow"), (rng.integers(-30, 31, (h, w)), rng.integers(-30, 31, (h, w)), "R=17 random jumps"),
                             (rng.normal(0, 2.5, (h, w)) + 6, rng.normal(0, 2.5, (h, w)) - 3, "R=17 noisy flow")):
            f = np.zeros((h, w), O.float2)
            f["x"], f["y"] = fx.astype(np.float32), fy.astype(np.float32)
            eq(S.c2f_refine(f, P0), O.c2f_refine(f, i1, i2, c1, c2, O.default_params(patch_r=17)), what)
    finally:
        S.set_params(None)


def test_end_to_end_crop(crop, crop_stages):
    import eppm_amd
    e = eppm_amd.EPPM()
    e.init(crop[0], crop[1], 120, 160)
    u, v = e.compute_flow()
    st = crop_stages
    eq(e.plane("nnf1", 2), st["nnf1_fill"], "pipeline NNF")
    eq(u, st["u"], "u")
    eq(v, st["v"], "v")
    epe = float(np.sqrt((u - st["u"]) ** 2 + (v - st["v"]) ** 2).mean())
    assert epe <= 1e-3     # north_star tolerance
    # set_data may be called repeatedly after one init (driver .cpp:159-168): same answer again
    e.set_data(crop[0], crop[1])
    u2, v2 = e.compute_flow()
    eq(u2, u, "second run u"); eq(v2, v, "second run v")


# ---------------------------------------------------------------------------------------------------
# whole path at other shapes / parameters
# ---------------------------------------------------------------------------------------------------
def _run_both(a, b, **params):
    import eppm_amd
    from oracle import oracle as O
    h, w, _ = a.shape
    e = eppm_amd.EPPM(params=eppm_amd.Params(**params) if params else None)
    e.init(a, b, h, w)
    u, v = e.compute_flow()
    ou, ov = O.compute_flow(a, b, O.default_params(**params) if params else None)
    return u, v, ou, ov


def test_odd_size_pair(frames):
    """123 x 157: no dimension is a multiple of 16, level dims are odd (61x78 -> 30x39), so the level-2
    decimation uses a non-exact ratio (real bilinear weights) and every tile kernel has ragged edges."""
    a, b = frames
    u, v, ou, ov = _run_both(a[100:223, 200:357].copy(), b[100:223, 200:357].copy())
    eq(u, ou, "u 157x123"); eq(v, ov, "v 157x123")


def test_patch_radius_17_and_other_parameters(crop):
    """BASELINE config 5 uses PATCH_R 17 (18x18 samples): the R=17 cooperative-sweep and tiled-refine
    instantiations; R=5 exercises the generic (non-templated) fallbacks; odd iteration / guess counts."""
    a, b = crop
    u, v, ou, ov = _run_both(a[:96, :128].copy(), b[:96, :128].copy(), patch_r=17)
    eq(u, ou, "u R=17"); eq(v, ov, "v R=17")
    u, v, ou, ov = _run_both(a[:64, :96].copy(), b[:64, :96].copy(), patch_r=5, num_iter=3, num_guess=5, seg_len=7, wmf_iters=4, search_range=11)
    eq(u, ou, "u R=5 generic"); eq(v, ov, "v R=5 generic")


@pytest.fixture(params=[-1, 3], ids=["sweeps_default", "sweeps_merged"])
def sweep_mode(request):
    """The library's own choice of the sweeps' form, and the merged speculative form forced from the first iteration (sweep_spec 3)."""
    import eppm_amd
    L = eppm_amd.lib()
    assert L.eppm_test_set_option(b"sweep_spec", request.param) == 0
    yield request.param
    L.eppm_test_set_option(b"sweep_spec", -1)


def test_tiny_and_degenerate_inputs(sweep_mode):
    """Smallest supported size, constant images (every patch cost ties at 0), and identical images."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (16, 20, 3), dtype=np.uint8); b = rng.integers(0, 256, (16, 20, 3), dtype=np.uint8)
    u, v, ou, ov = _run_both(a, b)
    eq(u, ou, "u 20x16"); eq(v, ov, "v 20x16")
    c = np.full((40, 48, 3), 99, np.uint8)
    u, v, ou, ov = _run_both(c, c)
    eq(u, ou, "u constant"); eq(v, ov, "v constant")
    black, white = np.zeros((40, 48, 3), np.uint8), np.full((40, 48, 3), 255, np.uint8)
    u, v, ou, ov = _run_both(black, white)       # all range weights flush to 0 except identical colours
    eq(u, ou, "u black/white"); eq(v, ov, "v black/white")


@pytest.mark.parametrize("h,w,params", [
    (24, 4000, {}),                                  # 1 x 250-pixel... quarter level 6 x 1000: 100 segments per row, 6 lines
    (4000, 24, {}),                                  # the transpose: column sweeps with 100 segments
    (17, 2051, dict(levels=2)),                      # odd sizes, two levels
    (2051, 17, dict(levels=1, seg_len=3)),           # single scale, 684 segments per column
    (64, 3000, dict(patch_r=17, num_iter=2)),        # radius 17 on a strip: the patch is taller than the image at every level
    (33, 1500, dict(propagation=1, num_iter=2)),     # jump flood on a strip
])
def test_extreme_aspect_ratios(h, w, params, sweep_mode):
    """Strips: many segments per line and few lines (and the transpose) -- grid and tile-mapping edge cases of the sweeps (in the
    library's own form and in the merged speculative form), the search and the refine; patches larger than the image."""
    import eppm_amd
    from oracle import oracle as O
    rng = np.random.default_rng([h, w])
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    b = np.roll(a, (1, -3), axis=(0, 1))
    b[::7] = rng.integers(0, 256, b[::7].shape, dtype=np.uint8)
    e = eppm_amd.EPPM(params=eppm_amd.Params(**params))
    e.init(a, b, h, w)
    u, v = e.compute_flow()
    e.close()
    ou, ov = O.compute_flow(a, b, O.default_params(**params))
    eq(u, ou, f"u {w}x{h} {params}"); eq(v, ov, f"v {w}x{h} {params}")


@pytest.mark.parametrize("params", [
    dict(seg_len=24),                      # the longest segments whose source tile (17 rows x 115 texels) still fits the LDS budget
    dict(seg_len=25),                      # one more: the sweeps gather their source samples
    dict(seg_len=64, patch_r=17),          # radius 17, long segments (gather path), 64-lane chains
    dict(seg_len=2),                       # shortest segments: two steps per chain
    dict(seg_len=200),                     # one segment per line at this size
    dict(num_guess=8, search_range=1),     # eight guesses, all at radius 1
    dict(num_guess=1, search_range=200),   # one guess anywhere in the image
    dict(wmf_iters=0, num_iter=1),
])
def test_sweep_and_search_parameter_extremes(frames, params):
    a, b = frames
    u, v, ou, ov = _run_both(a[40:231, 100:421].copy(), b[40:231, 100:421].copy(), **params)
    eq(u, ou, f"u {params}"); eq(v, ov, f"v {params}")


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("params,region", [
    (dict(), (slice(40, 231), slice(100, 421))),
    (dict(seg_len=16, num_iter=5), (slice(40, 231), slice(100, 421))),        # the longest segments phase B takes (16 lanes per chain)
    (dict(seg_len=17, num_iter=4), (slice(40, 231), slice(100, 421))),        # one more: classic form whatever the switch says
    (dict(seg_len=2, num_iter=4), (slice(100, 223), slice(200, 357))),        # two steps per chain, odd size
    (dict(patch_r=17, num_iter=4), (slice(0, 192), slice(0, 256))),           # radius 17: 64 lanes per chain
    (dict(patch_r=5, num_iter=3), (slice(0, 128), slice(0, 192))),            # no instantiation for this radius: classic fallback
    (dict(levels=1, num_iter=4), (slice(180, 300), slice(240, 400))),         # single scale
])
def test_speculative_sweeps_forced_on_and_off(frames, params, region, mode):
    """The speculative two-launch sweeps (phase A evaluates every pixel's rejection-path candidate in parallel, phase B walks
    the chains) against the classic dependent-step kernel: forced for EVERY iteration (mode 1: also the first ones, where most
    steps follow an accepted candidate and phase B evaluates; with the work list -- phase B walks only the chains on which phase A
    found a candidate that would be accepted -- and, mode 2, without it; mode 3: the merged form, ONE phase A for the four sweeps of an
    iteration and four in-place launches over their lists, also from the first iteration, where nearly every chain is listed and the
    later sweeps' lists come mostly from the earlier sweeps' accepted candidates) and forced off (mode 0), all == the oracle bit for bit."""
    import eppm_amd
    a, b = frames
    L = eppm_amd.lib()
    try:
        assert L.eppm_test_set_option(b"sweep_spec", mode) == 0
        u, v, ou, ov = _run_both(a[region].copy(), b[region].copy(), **params)
    finally:
        L.eppm_test_set_option(b"sweep_spec", -1)
    eq(u, ou, f"u {params} sweep_spec={mode}"); eq(v, ov, f"v {params} sweep_spec={mode}")


def test_bundled_pair_full_size(frames):
    """BASELINE config 1: frame10/frame11 (640x480), default parameters.  HIP flow == oracle flow bit for bit
    (EPE 0 <= 1e-3 px), and the oracle's flow matches the committed sha256."""
    import hashlib, json, os
    from conftest import GOLDEN
    a, b = frames
    u, v, ou, ov = _run_both(a, b)
    epe = float(np.sqrt((u - ou) ** 2 + (v - ov) ** 2).mean())
    assert epe <= 1e-3
    eq(u, ou, "u 640x480"); eq(v, ov, "v 640x480")
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    assert hashlib.sha256(u.tobytes() + v.tobytes()).hexdigest() == man["oracle_flow_640x480_sha256"]


def test_sintel_shape_properties():
    """BASELINE config 2 at full size (1024x436): size-independent properties instead of the slow oracle --
    determinism across runs and contexts, host/device entry points agree, integer translation is recovered."""
    import eppm_amd
    from eppm_amd import synth
    h, w = 436, 1024
    a, b, gu, gv = synth.make_pair(h, w, seed=1234)
    e1 = eppm_amd.EPPM(); e1.init(a, b, h, w)
    u1, v1 = e1.compute_flow()
    e1.set_data(a, b)
    u2, v2 = e1.compute_flow()
    e2 = eppm_amd.EPPM(); e2.init(h, w); e2.set_data(a, b)
    u3, v3 = e2.compute_flow()
    eq(u1, u2, "same context twice"); eq(v1, v2, "same context twice")
    eq(u1, u3, "second context"); eq(v1, v3, "second context")
    assert np.isfinite(u1).all() and np.abs(u1).max() < 200
    # a pure translation by (8, -4): the interior flow is exactly that vector almost everywhere
    sh = np.roll(a, (-4, 8), axis=(0, 1))
    e2.set_data(a, sh)
    tu, tv = e2.compute_flow()
    inner = (slice(40, h - 40), slice(40, w - 40))
    good = (np.abs(tu[inner] - 8) < 0.5) & (np.abs(tv[inner] + 4) < 0.5)
    assert good.mean() > 0.97, good.mean()


def test_c_class_cli_matches_library(frames, tmp_path):
    """tools/runeppm (the reference's main.cpp I/O contract through the drop-in C++ class) writes the same
    .flo as the library called from Python."""
    import os, subprocess
    import eppm_amd
    from conftest import GOLDEN
    exe = os.path.join(os.path.dirname(eppm_amd.lib_path()), "runeppm")
    out = str(tmp_path / "flow.flo")
    subprocess.check_call([exe, os.path.join(GOLDEN, "frame10.ppm"), os.path.join(GOLDEN, "frame11.ppm"), out])
    fu, fv = eppm_amd.io.load_flo(out)
    a, b = frames
    e = eppm_amd.EPPM(); e.init(a, b, 480, 640)
    u, v = e.compute_flow()
    eq(fu, u, "CLI u"); eq(fv, v, "CLI v")


def test_cli_options(frames, tmp_path):
    """runeppm --levels/--patch-r/--iters/--propagation/--seed reach the engine (same .flo as the library with those
    parameters); --size runs a synthetic translated pair and recovers the translation; --pairs/--gpus stream pairs."""
    import os, re, subprocess
    import eppm_amd
    from conftest import GOLDEN
    exe = os.path.join(os.path.dirname(eppm_amd.lib_path()), "runeppm")
    out = str(tmp_path / "f.flo")
    txt = subprocess.check_output([exe, "--levels", "2", "--patch-r", "5", "--iters", "2", "--propagation", "1", "--seed", "77", "--pairs", "2",
                                   os.path.join(GOLDEN, "frame10.ppm"), os.path.join(GOLDEN, "frame11.ppm"), out], text=True)
    assert "Mflow-vectors/s" in txt
    fu, fv = eppm_amd.io.load_flo(out)
    a, b = frames
    e = eppm_amd.EPPM(params=eppm_amd.Params(levels=2, patch_r=5, num_iter=2, propagation=1, seed=77)); e.init(a, b, 480, 640)
    u, v = e.compute_flow()
    eq(fu, u, "CLI u with options"); eq(fv, v, "CLI v with options")
    txt = subprocess.check_output([exe, "--size", "320x200", "--pairs", "7", "--batch", "3", "--out", out], text=True)      # batch contexts: 3 + 3 + 1
    assert "3 pair(s) per launch" in txt and "Mflow-vectors/s" in txt, txt
    txt = subprocess.check_output([exe, "--size", "320x200", "--pairs", "3", "--out", out], text=True)
    m = re.search(r"EPE ([0-9.]+) px", txt)
    assert m and float(m.group(1)) < 0.5, txt
    assert subprocess.run([exe, "--levels"], capture_output=True).returncode == 2
    assert subprocess.run([exe, "--levels", "99", "--size", "64x64"], capture_output=True).returncode == 1


def test_jump_flood_propagation(S, O, L1, crop):
    """Optional mode (eppm_params.propagation = 1): the reference's baoJumpPropagate (disabled there, kernel.cu:1813)."""
    i1, i2, c1, c2 = L1
    h, w = i1.shape
    P = S.PlaneSet(i1, i2, c1, c2)
    onnf, _ = O.gen_rand_field(w, h)
    ocost = O.cost_field(onnf, i1, i2, c1, c2)
    cost, nnf = S.pm_jump_propagate(ocost, onnf, P)
    oc, on = O.jump_propagate(ocost, onnf, i1, i2, c1, c2)
    eq(nnf, on, "NNF after jump flood"); eq(cost, oc, "cost after jump flood")
    cost, nnf = S.pm_jump_propagate(cost, nnf, P)              # second round: many candidates equal the own match
    oc, on = O.jump_propagate(oc, on, i1, i2, c1, c2)
    eq(nnf, on, "NNF after 2nd jump flood"); eq(cost, oc, "cost after 2nd jump flood")
    a, b = crop
    u, v, ou, ov = _run_both(a, b, propagation=1)
    eq(u, ou, "u jump flood"); eq(v, ov, "v jump flood")


def test_four_neighbour_propagation(S, O, L1, crop):
    """Optional mode (eppm_params.propagation = 2): the reference's baoParallelPropagate, ten launches per iteration
    (disabled there, kernel.cu:1804-1809)."""
    i1, i2, c1, c2 = L1
    h, w = i1.shape
    P = S.PlaneSet(i1, i2, c1, c2)
    onnf, _ = O.gen_rand_field(w, h)
    ocost = O.cost_field(onnf, i1, i2, c1, c2)
    cost, nnf, oc, on = ocost, onnf, ocost, onnf
    for launch in range(3):                                     # later launches: many candidates equal the own match
        cost, nnf = S.pm_parallel_propagate(cost, nnf, P)
        oc, on = O.parallel_propagate(oc, on, i1, i2, c1, c2)
        eq(nnf, on, "NNF after 4-neighbour launch %d" % launch); eq(cost, oc, "cost after 4-neighbour launch %d" % launch)
    a, b = crop
    u, v, ou, ov = _run_both(a, b, propagation=2)
    eq(u, ou, "u 4-neighbour"); eq(v, ov, "v 4-neighbour")


@pytest.mark.parametrize("levels", [1, 2, 4])
def test_pyramid_depth(crop, levels):
    """PYR_MAX_DEPTH (defs.h:31) as a run-time parameter: PatchMatch at level levels-1, levels-1 C2F steps.
    levels = 1 is the single-scale run of BASELINE.json's first configuration."""
    import eppm_amd
    a, b = crop
    u, v, ou, ov = _run_both(a, b, levels=levels, num_iter=3)
    eq(u, ou, "u levels=%d" % levels); eq(v, ov, "v levels=%d" % levels)
    e = eppm_amd.EPPM(params=eppm_amd.Params(levels=levels)); e.init(120, 160)
    assert len(e.level_dims()) == levels
    with pytest.raises(eppm_amd.EppmError):
        eppm_amd.EPPM(params=eppm_amd.Params(levels=9)).init(120, 160)


def test_pipelined_host_boundary(crop, crop_stages):
    """eppm_compute_begin / eppm_compute_end: three contexts kept in flight by one host thread return, for every pair, the
    flow of the synchronous call; end without begin is an error."""
    import eppm_amd
    from eppm_amd import shard
    a, b = crop
    st = crop_stages
    pairs = [(a, b), (b, a), (a, a), (a, b), (b, a), (a, b), (a, b)]
    ref = eppm_amd.EPPM(); ref.init(120, 160)
    want = shard.run_pairs(ref, pairs, range(len(pairs)))
    eq(want[0][0], st["u"], "sync u"); eq(want[0][1], st["v"], "sync v")
    engs = []
    for _ in range(3):
        e = eppm_amd.EPPM(); e.init(120, 160); engs.append(e)
    got = shard.run_pairs_pipelined(engs, pairs, range(len(pairs)))
    assert sorted(got) == list(range(len(pairs)))
    for i in got:
        eq(got[i][0], want[i][0], "pipelined u %d" % i); eq(got[i][1], want[i][1], "pipelined v %d" % i)
    with pytest.raises(eppm_amd.EppmError):
        engs[0].compute_flow_end()


def test_device_entry_points_and_streams(crop, crop_stages):
    """eppm_set_images_device / eppm_compute_device (inputs resident in HBM, any pitch) give the same flow as the
    host-pointer entry points; contexts on separate streams can be in flight together."""
    import ctypes as C
    import eppm_amd
    from eppm_amd import stages as S
    from oracle import oracle as O
    a, b = crop
    h, w = 120, 160
    st = crop_stages
    for pitched in (False, True):
        d1, d2 = S.Dev(O.rgb2rgba(a), pitched=pitched), S.Dev(O.rgb2rgba(b), pitched=pitched)
        out = S.Dev(shape=(h, w), dtype=eppm_amd.api.float2)
        engs = [eppm_amd.EPPM() for _ in range(3)]
        for e in engs:
            e.init(h, w)
        for e in engs:                                   # three pairs in flight, one stream each
            e.set_data_device(d1.ptr.value, d2.ptr.value, d1.pitch)
            e.compute_flow_device(out.ptr.value if e is engs[0] else None)
        for e in engs:
            e.synchronize()
        f0 = out.get()
        eq(f0["x"].copy(), st["u"], "device entry u"); eq(f0["y"].copy(), st["v"], "device entry v")
        for e in engs[1:]:
            f = e.plane("flow", 0)
            eq(f["x"].copy(), st["u"], "in-flight context u"); eq(f["y"].copy(), st["v"], "in-flight context v")


def test_error_paths_on_gpu():
    import ctypes as C
    import eppm_amd
    L = eppm_amd.lib()
    e = eppm_amd.EPPM()
    e.init(64, 64)
    with pytest.raises(eppm_amd.EppmError):
        e.compute_flow()                                  # no images yet: EPPM_ERR_STATE
    with pytest.raises(eppm_amd.EppmError):
        e.set_data(np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10, 3), np.uint8))
    with pytest.raises(eppm_amd.EppmError):
        e.plane("flow", 7)
    ctx = C.c_void_p()
    assert L.eppm_create(C.byref(ctx), 64, 64, 99, None) == 2    # no such device: EPPM_ERR_HIP, no exit()
    e.init(32, 48)                                        # re-init releases the old buffers (the reference leaks them)
    z = np.zeros((32, 48, 3), np.uint8)
    e.set_data(z, z)
    u, v = e.compute_flow()
    assert u.shape == (32, 48) and np.isfinite(u).all()
    # batch API: argument errors, and an allocation that cannot succeed (4096 slabs of a 4K pair) fails cleanly
    assert L.eppm_create_batch(C.byref(ctx), 64, 64, 0, None, 0) == 1
    assert L.eppm_create_batch(C.byref(ctx), 2160, 3840, 0, None, 4096) == 2 and b"hipMalloc" in L.eppm_last_error()
    B = eppm_amd.EPPMBatch(32, 48, 2)
    with pytest.raises(eppm_amd.EppmError):
        B.set_data([(z, z), (z, z), (z, z)])             # 3 pairs into a 2-pair context
    with pytest.raises(eppm_amd.EppmError):
        B.compute_flow_device()                           # no images yet
    B.set_data([(z, z)])
    out = B.compute_flow()
    assert len(out) == 1 and np.array_equal(out[0][0].view(np.uint32), u.view(np.uint32))
    with pytest.raises(eppm_amd.EppmError):
        B.plane(5, "flow", 0)
    B.close()


def test_contexts_on_host_threads(crop, crop_stages):
    """One context per host thread (SURVEY 8b threading contract): four threads, each its own context, same answer."""
    import threading
    import eppm_amd
    a, b = crop
    st = crop_stages
    results, errors = [None] * 4, []

    def work(i):
        try:
            e = eppm_amd.EPPM()
            e.init(120, 160)
            for _ in range(3):
                e.set_data(a, b)
                results[i] = e.compute_flow()
        except Exception as ex:       # noqa: BLE001
            errors.append(ex)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for u, v in results:
        eq(u, st["u"], "thread u"); eq(v, st["v"], "thread v")


def test_no_device_memory_leak_over_create_destroy(crop):
    """init/destroy cycles and repeated set_data/compute_flow leave the free device memory where it was
    (the reference leaks its buffers when init is called twice, driver .cpp:112-157)."""
    import ctypes as C
    import eppm_amd
    from eppm_amd._lib import lib, check
    a, b = crop
    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        check(lib().eppm_device_synchronize(), "sync")
        check(lib().eppm_device_mem_info(C.byref(f), C.byref(t)), "mem_info")
        return f.value
    e = eppm_amd.EPPM(); e.init(a, b, 120, 160); e.compute_flow(); e.close()      # warm up allocator pools
    before = free_bytes()
    for _ in range(25):
        e = eppm_amd.EPPM()
        e.init(120, 160)
        e.enable_stage_timing(True)
        for _ in range(2):
            e.set_data(a, b)
            e.compute_flow()
        e.init(96, 128)                       # re-init on a live object
        e.close()
    after = free_bytes()
    assert abs(before - after) < 8 << 20, (before, after)
    # a destroyed context's slab, staging buffers and stream are kept for the next context of that size (allocation is inside the window
    # the reference's demo times); eppm_release_cached_memory hands them back: a 1024x436 slab is 95 MB
    e = eppm_amd.EPPM(); e.init(436, 1024); e.close()
    held = free_bytes()
    check(lib().eppm_release_cached_memory(), "release")
    assert free_bytes() - held > 64 << 20, (held, free_bytes())
    # ... and the generator tables no context uses any more (block start states + the numbers of a run drawn ahead: 7 MB at 1024x436,
    # 31 MB at 1920x1080; kept per geometry while idle): several geometries come and go, the release returns to the level before them
    level = free_bytes()
    for hh, ww in ((436, 1024), (540, 960), (600, 800), (1080, 1920)):
        e = eppm_amd.EPPM(); e.init(hh, ww); e.close()
    assert level - free_bytes() > 32 << 20                       # (held: slabs and tables)
    check(lib().eppm_release_cached_memory(), "release")
    assert abs(free_bytes() - level) < 8 << 20, (level, free_bytes())
    e = eppm_amd.EPPM(); e.init(a, b, 120, 160); u1, v1 = e.compute_flow(); e.close()       # and everything still works afterwards
    e = eppm_amd.EPPM(); e.init(a, b, 120, 160); u2, v2 = e.compute_flow(); e.close()
    assert np.array_equal(u1, u2) and np.array_equal(v1, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3])
def test_search_draws_in_context_when_no_table(frames, batch):
    """A context whose geometry would need more than 512 MB of numbers drawn ahead searches with the drawing wave and per-pair generator
    states instead (k_pm_random_search<.., TAB = false>, the default of rounds 1-3): forced here with the "rand_table" switch, single
    and batch context, == the oracle bit for bit, and == a context with the table."""
    import eppm_amd
    from oracle import oracle as O
    a, b = frames
    region = (slice(60, 300), slice(100, 500))
    a, b = np.ascontiguousarray(a[region]), np.ascontiguousarray(b[region])
    L = eppm_amd.lib()
    ou, ov = O.compute_flow(a, b)
    try:
        assert L.eppm_test_set_option(b"rand_table", 0) == 0
        if batch == 1:
            e = eppm_amd.EPPM(); e.init(*a.shape[:2]); e.set_data(a, b); flows = [e.compute_flow()]; e.close()
        else:
            e = eppm_amd.EPPMBatch(a.shape[0], a.shape[1], batch); e.set_data([(a, b)] * batch); flows = e.compute_flow(); e.close()
    finally:
        L.eppm_test_set_option(b"rand_table", 1)
    for u, v in flows:
        eq(u, ou, "u, streaming search"); eq(v, ov, "v, streaming search")
