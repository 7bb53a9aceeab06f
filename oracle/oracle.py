"""ctypes front end of the CPU oracle (oracle/eppm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (eppm_amd/) never imports this module.
Parity status: "parity unpinned" against the CUDA original (see eppm_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libeppm_oracle.so")
_REFIO_PATH = os.path.join(_HERE, "_ref", "libeppm_refio.so")
_RUNREF_PATH = os.path.join(_HERE, "_ref", "runeppm_ref")

uchar4 = np.dtype([("x", "u1"), ("y", "u1"), ("z", "u1"), ("w", "u1")])
short2 = np.dtype([("x", "i2"), ("y", "i2")])
float2 = np.dtype([("x", "f4"), ("y", "f4")])


class Params(C.Structure):
    _fields_ = [("patch_r", C.c_int), ("num_iter", C.c_int), ("search_range", C.c_int), ("num_guess", C.c_int),
                ("seg_len", C.c_int), ("wmf_iters", C.c_int), ("seed", C.c_ulonglong), ("dump_stages", C.c_int),
                ("propagation", C.c_int), ("levels", C.c_int)]


class Xorwow(C.Structure):
    _fields_ = [("v", C.c_uint32 * 5), ("d", C.c_uint32)]


class Dump(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("arrH", C.c_int * 8), ("arrW", C.c_int * 8),
                ("img1", C.c_void_p * 8), ("img2", C.c_void_p * 8), ("cen1", C.c_void_p * 8), ("cen2", C.c_void_p * 8),
                ("nnf1_pm", C.c_void_p), ("nnf2_pm", C.c_void_p), ("cost1_pm", C.c_void_p), ("cost2_pm", C.c_void_p),
                ("nnf1_lr", C.c_void_p), ("nnf1_out", C.c_void_p), ("nnf1_wmf", C.c_void_p), ("nnf1_fill", C.c_void_p),
                ("flow", C.c_void_p * 8), ("flow_c2f", C.c_void_p * 8)]


def build(force=False):
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "eppm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "_build/libeppm_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REFIO_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    # the reference's main.cpp, unmodified, on the drop-in headers + libeppm_hip.so (needs the HIP library built first)
    hip_lib = os.path.join(_HERE, "..", "eppm_amd", "lib", "libeppm_hip.so")
    if os.path.isdir("/root/reference") and os.path.exists(hip_lib) and \
            (force or not os.path.exists(_RUNREF_PATH) or os.path.getmtime(_RUNREF_PATH) < os.path.getmtime(hip_lib)):
        subprocess.check_call(["make", "-C", _HERE, "runeppm_ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fast_exp.restype = C.c_float
        _lib.orc_fast_exp.argtypes = [C.c_float]
        _lib.orc_patch_dist.restype = C.c_float
        _lib.orc_patch_dist_planefit.restype = C.c_float
        _lib.orc_xorwow_next.restype = C.c_uint32
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def refio():
    """The reference's own host I/O code (oracle/_ref), or None when it was not built."""
    if not os.path.exists(_REFIO_PATH):
        return None
    return C.CDLL(_REFIO_PATH)


def runeppm_ref():
    """Path of the reference's own main.cpp built unmodified on the drop-in boundary (oracle/_ref), or None."""
    return _RUNREF_PATH if os.path.exists(_RUNREF_PATH) else None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def fast_exp(x):
    x = np.asarray(x, np.float32)
    out = np.empty_like(x)
    f = lib().orc_fast_exp
    for i, v in enumerate(x.ravel()):
        out.ravel()[i] = f(float(v))
    return out


def pm_luts(patch_r=9):
    gs = np.zeros(patch_r + 1, np.float32)
    cn = np.zeros(9, np.float32)
    lib().orc_pm_luts(patch_r, _p(gs), _p(cn))
    return gs, cn


def wmf_lut():
    g = np.zeros(5, np.float32)
    lib().orc_wmf_lut(_p(g))
    return g


def blf_lut():
    g = np.zeros(11, np.float32)
    lib().orc_blf_lut(_p(g))
    return g


def xorwow_stream(seed, subsequence, n, skip=0):
    s = Xorwow()
    lib().orc_xorwow_init(C.byref(s), C.c_ulonglong(seed), C.c_ulonglong(subsequence))
    if skip:
        lib().orc_xorwow_skip(C.byref(s), C.c_ulonglong(skip))
    out = np.empty(n, np.uint32)
    nxt = lib().orc_xorwow_next
    for i in range(n):
        out[i] = nxt(C.byref(s))
    return out


def xorwow_state(seed, subsequence, skip=0):
    s = Xorwow()
    lib().orc_xorwow_init(C.byref(s), C.c_ulonglong(seed), C.c_ulonglong(subsequence))
    if skip:
        lib().orc_xorwow_skip(C.byref(s), C.c_ulonglong(skip))
    return np.array(list(s.v) + [s.d], np.uint32)


def pyr_init_dim(h, w, max_depth=3, ratio=0.5):
    ah = (C.c_int * 8)()
    aw = (C.c_int * 8)()
    n = lib().orc_pyr_init_dim(ah, aw, h, w, max_depth, C.c_float(ratio))
    return [ah[i] for i in range(n)], [aw[i] for i in range(n)]


def rgb2rgba(rgb):
    h, w, _ = rgb.shape
    out = np.zeros((h, w), uchar4)
    lib().orc_rgb2rgba(_p(out), _p(np.ascontiguousarray(rgb, np.uint8)), h, w)
    return out


def gauss_filter_rgba(img, sigma, radius):
    h, w = img.shape
    out = np.zeros((h, w), uchar4)
    lib().orc_gauss_filter_rgba(_p(out), _p(np.ascontiguousarray(img)), h, w, C.c_float(sigma), radius)
    return out


def resize_rgba(img, out_h, out_w, ratio):
    h, w = img.shape
    out = np.zeros((out_h, out_w), uchar4)
    lib().orc_resize_rgba(_p(out), out_h, out_w, _p(np.ascontiguousarray(img)), h, w, C.c_float(ratio))
    return out


def census(img):
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_census(_p(out), _p(np.ascontiguousarray(img)), h, w)
    return out


def prepare(raw_rgba, n_levels=3):
    h, w = raw_rgba.shape
    ah, aw = pyr_init_dim(h, w, n_levels)
    imgs = [np.zeros((ah[i], aw[i]), uchar4) for i in range(n_levels)]
    cens = [np.zeros((ah[i], aw[i]), np.uint8) for i in range(n_levels)]
    ip = (C.c_void_p * n_levels)(*[a.ctypes.data for a in imgs])
    cp = (C.c_void_p * n_levels)(*[a.ctypes.data for a in cens])
    lib().orc_prepare(ip, cp, _p(np.ascontiguousarray(raw_rgba)), (C.c_int * n_levels)(*ah), (C.c_int * n_levels)(*aw), n_levels)
    return imgs, cens


def _planes(img1, img2, c1, c2):
    h, w = img1.shape
    return (_p(np.ascontiguousarray(img1)), _p(np.ascontiguousarray(img2)), _p(np.ascontiguousarray(c1)),
            _p(np.ascontiguousarray(c2)), w, h)


def patch_dist(img1, img2, c1, c2, x1, y1, x2, y2, patch_r=9, planefit=False):
    gs, cn = pm_luts(patch_r)
    fn = lib().orc_patch_dist_planefit if planefit else lib().orc_patch_dist
    return fn(*_planes(img1, img2, c1, c2), patch_r, _p(gs), _p(cn), x1, y1, x2, y2)


def gen_rand_field(w, h, seed=1234):
    gx, gy = (w + 15) // 16, (h + 15) // 16
    states = np.zeros((gx * gy, 6), np.uint32)
    nnf = np.zeros((h, w), short2)
    lib().orc_gen_rand_field(_p(states), _p(nnf), w, h, C.c_ulonglong(seed))
    return nnf, states


def cost_field(nnf, img1, img2, c1, c2, params=None):
    params = params or default_params()
    h, w = img1.shape
    cost = np.zeros((h, w), np.float32)
    lib().orc_cost_field(_p(cost), _p(np.ascontiguousarray(nnf)), *_planes(img1, img2, c1, c2), C.byref(params))
    return cost


def seg_propagate_dir(cost, nnf, img1, img2, c1, c2, direction, params=None):
    """In place on copies; returns (cost, nnf)."""
    params = params or default_params()
    cost = np.ascontiguousarray(cost).copy()
    nnf = np.ascontiguousarray(nnf).copy()
    lib().orc_seg_propagate_dir(_p(cost), _p(nnf), *_planes(img1, img2, c1, c2), C.byref(params), direction)
    return cost, nnf


def jump_propagate(cost, nnf, img1, img2, c1, c2, params=None):
    params = params or default_params()
    cost = np.ascontiguousarray(cost).copy()
    nnf = np.ascontiguousarray(nnf).copy()
    lib().orc_jump_propagate(_p(cost), _p(nnf), *_planes(img1, img2, c1, c2), C.byref(params))
    return cost, nnf


def parallel_propagate(cost, nnf, img1, img2, c1, c2, params=None):
    params = params or default_params()
    cost = np.ascontiguousarray(cost).copy()
    nnf = np.ascontiguousarray(nnf).copy()
    lib().orc_parallel_propagate(_p(cost), _p(nnf), *_planes(img1, img2, c1, c2), C.byref(params))
    return cost, nnf


def random_search(states, cost, nnf, img1, img2, c1, c2, params=None):
    params = params or default_params()
    cost = np.ascontiguousarray(cost).copy()
    nnf = np.ascontiguousarray(nnf).copy()
    states = np.ascontiguousarray(states).copy()
    lib().orc_random_search(_p(states), _p(cost), _p(nnf), *_planes(img1, img2, c1, c2), C.byref(params))
    return states, cost, nnf


def patchmatch(img1, img2, c1, c2, params=None, iters_done=-1):
    params = params or default_params()
    h, w = img1.shape
    nnf = np.zeros((h, w), short2)
    cost = np.zeros((h, w), np.float32)
    lib().orc_patchmatch(_p(nnf), _p(cost), *_planes(img1, img2, c1, c2), C.byref(params), iters_done)
    return nnf, cost


def left_right_check(nnf1, cost1, nnf2, cost2):
    h, w = nnf1.shape
    a, b, c, d = [np.ascontiguousarray(t).copy() for t in (nnf1, cost1, nnf2, cost2)]
    lib().orc_left_right_check(_p(a), _p(b), _p(c), _p(d), w, h)
    return a, b, c, d


def outlier_removal(nnf, cost):
    h, w = nnf.shape
    a, b = np.ascontiguousarray(nnf).copy(), np.ascontiguousarray(cost).copy()
    lib().orc_outlier_removal(_p(a), _p(b), w, h)
    return a, b


def weighted_median(nnf, img, num_iter=20, only_occlusion=True):
    h, w = nnf.shape
    a = np.ascontiguousarray(nnf).copy()
    lib().orc_weighted_median(_p(a), _p(np.ascontiguousarray(img)), w, h, num_iter, int(only_occlusion))
    return a


def fill_holes(nnf, img):
    h, w = nnf.shape
    a = np.ascontiguousarray(nnf).copy()
    lib().orc_fill_holes(_p(a), _p(np.ascontiguousarray(img)), w, h)
    return a


def nnf2flow(nnf):
    h, w = nnf.shape
    f = np.zeros((h, w), float2)
    lib().orc_nnf2flow(_p(f), _p(np.ascontiguousarray(nnf)), w, h)
    return f


def resize_flow(flow, out_h, out_w, ratio=2.0):
    h, w = flow.shape
    out = np.zeros((out_h, out_w), float2)
    lib().orc_resize_flow(_p(out), out_h, out_w, _p(np.ascontiguousarray(flow)), h, w, C.c_float(ratio))
    return out


def mul_scalar(flow, s):
    h, w = flow.shape
    a = np.ascontiguousarray(flow).copy()
    lib().orc_mul_scalar(_p(a), C.c_float(s), h, w)
    return a


def c2f_refine(flow, img1, img2, c1, c2, params=None):
    params = params or default_params()
    a = np.ascontiguousarray(flow).copy()
    lib().orc_c2f_refine(_p(a), *_planes(img1, img2, c1, c2), C.byref(params))
    return a


def flow_smoothing(flow, img):
    h, w = flow.shape
    a = np.ascontiguousarray(flow).copy()
    lib().orc_flow_smoothing(_p(a), _p(np.ascontiguousarray(img)), w, h)
    return a


def compute_flow(rgb1, rgb2, params=None, dump=False):
    """Whole path (set_data + compute_flow).  rgb: (h,w,3) uint8.  Returns (u, v[, stages])."""
    params = params or default_params()
    h, w, _ = rgb1.shape
    u = np.zeros((h, w), np.float32)
    v = np.zeros((h, w), np.float32)
    d = Dump()
    lib().orc_compute_flow(_p(np.ascontiguousarray(rgb1, np.uint8)), _p(np.ascontiguousarray(rgb2, np.uint8)), h, w,
                           C.byref(params), _p(u), _p(v), C.byref(d) if dump else None)
    if not dump:
        return u, v

    def grab(ptr, shape, dt):
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        return np.frombuffer(C.string_at(ptr, n), dtype=dt).reshape(shape).copy()

    st = {"arrH": list(d.arrH)[:d.n_levels], "arrW": list(d.arrW)[:d.n_levels]}
    L = d.n_levels - 1
    for i in range(d.n_levels):
        shp = (d.arrH[i], d.arrW[i])
        st[f"img1_L{i}"] = grab(d.img1[i], shp, uchar4)
        st[f"img2_L{i}"] = grab(d.img2[i], shp, uchar4)
        st[f"cen1_L{i}"] = grab(d.cen1[i], shp, np.uint8)
        st[f"cen2_L{i}"] = grab(d.cen2[i], shp, np.uint8)
        st[f"flow_L{i}"] = grab(d.flow[i], shp, float2)
        if i < L:
            st[f"flow_c2f_L{i}"] = grab(d.flow_c2f[i], shp, float2)
    shp = (d.arrH[L], d.arrW[L])
    for k in ("nnf1_pm", "nnf2_pm", "nnf1_lr", "nnf1_out", "nnf1_wmf", "nnf1_fill"):
        st[k] = grab(getattr(d, k), shp, short2)
    for k in ("cost1_pm", "cost2_pm"):
        st[k] = grab(getattr(d, k), shp, np.float32)
    lib().orc_free_dump(C.byref(d))
    return u, v, st


def set_variant(sweep_order=0, post_inplace=0, exp_mode=0, seed_variant=0):
    """Select another legal reading of the reference's racy / unspecified parts (eppm_oracle.c: orc_set_variant); call with no
    arguments to return to the lockstep oracle.  Used only by tools/parity_envelope.py and its test."""
    lib().orc_set_variant(int(sweep_order), int(post_inplace), int(exp_mode), int(seed_variant))


def set_tol_variant(mode=0, scope=0):
    """Tolerance-arithmetic variants of the patch term (eppm_oracle.c: orc_set_tol_variant); (0, 0) = the lockstep oracle.
    Used only by tools/tolerance_envelope.py and its test."""
    lib().orc_set_tol_variant(int(mode), int(scope))


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def flow_to_color(flow, max_disp_x=20.0, max_disp_y=20.0):
    """flow: (h,w) float2 -> (h,w) uchar4 {R,G,B,0} (basic/bao_basic_cuda.cuh:776-845)."""
    h, w = flow.shape
    out = np.zeros((h, w), uchar4)
    lib().orc_flow_to_color(_p(out), _p(np.ascontiguousarray(flow)), h, w, C.c_float(max_disp_x), C.c_float(max_disp_y))
    return out
