"""The reference's stage launchers (driver .cpp:40-62) callable on numpy planes.

Every function uploads its inputs, calls the ``baoCuda*`` / ``eppm_pm_*`` entry point of the C ABI on
device buffers, and downloads the result: the parity tests read like calls of the reference's own
launchers.  Structured dtypes: uchar4 / short2 / float2 as in eppm_amd.api.
"""
import ctypes as C

import numpy as np

from ._lib import CParams, check, check_launcher, lib
from .api import float2, short2, uchar4


class Dev:
    """A pitched or linear device buffer holding a 2-D plane."""

    def __init__(self, arr=None, shape=None, dtype=None, pitched=False):
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            shape, dtype = arr.shape, arr.dtype
        self.h, self.w = shape
        self.dtype = np.dtype(dtype)
        row = self.w * self.dtype.itemsize
        self.ptr = C.c_void_p()
        if pitched:
            pitch = C.c_size_t()
            check(lib().eppm_malloc_pitched(C.byref(self.ptr), C.byref(pitch), C.c_size_t(row), C.c_size_t(self.h)), "malloc_pitched")
            self.pitch = pitch.value
        else:
            check(lib().eppm_malloc_device(C.byref(self.ptr), C.c_size_t(row * self.h)), "malloc")
            self.pitch = row
        if arr is not None:
            self.put(arr)

    def put(self, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        row = self.w * self.dtype.itemsize
        check(lib().eppm_memcpy2d_h2d(self.ptr, C.c_size_t(self.pitch), arr.ctypes.data_as(C.c_void_p), C.c_size_t(row),
                                      C.c_size_t(row), C.c_size_t(self.h)), "h2d")

    def get(self):
        out = np.empty((self.h, self.w), self.dtype)
        row = self.w * self.dtype.itemsize
        check(lib().eppm_memcpy2d_d2h(out.ctypes.data_as(C.c_void_p), C.c_size_t(row), self.ptr, C.c_size_t(self.pitch),
                                      C.c_size_t(row), C.c_size_t(self.h)), "d2h")
        return out

    def free(self):
        if self.ptr:
            lib().eppm_free_device(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def set_params(params=None):
    check(lib().eppm_set_launcher_params(C.byref(params) if params is not None else None), "eppm_set_launcher_params")


def _sz(v):
    return C.c_size_t(v)


def probe_fast_exp(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    check(lib().eppm_probe_fast_exp(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size), "probe_fast_exp")
    return y


def probe_div_const(x, which):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    check(lib().eppm_probe_div_const(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size, which), "probe_div_const")
    return y


def probe_delta_table(d, which):
    """the table form of a range term at the distances d (test library; include/eppm_test.h)"""
    x = np.ascontiguousarray(d, np.float32)
    y = np.empty_like(x)
    check(lib().eppm_probe_delta_table(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size, which), "probe_delta_table")
    return y


def gauss_filter_rgba(img, sigma, radius):
    h, w = img.shape
    a, b = Dev(img, pitched=True), Dev(shape=(h, w), dtype=uchar4, pitched=True)
    check(lib().eppm_gauss_filter_rgba(b.ptr, a.ptr, _sz(a.pitch), h, w, C.c_float(sigma), radius), "gauss")
    return b.get()


def resize_rgba(img, out_h, out_w, ratio):
    h, w = img.shape
    a, b = Dev(img, pitched=True), Dev(shape=(out_h, out_w), dtype=uchar4, pitched=True)
    check(lib().eppm_resize_rgba(b.ptr, _sz(b.pitch), out_h, out_w, a.ptr, _sz(a.pitch), h, w, C.c_float(ratio)), "resize_rgba")
    return b.get()


def census_transform(img1, img2):
    """baoCudaCensusTransform"""
    h, w = img1.shape
    a, b = Dev(img1, pitched=True), Dev(img2, pitched=True)
    c1, c2 = Dev(shape=(h, w), dtype=np.uint8, pitched=True), Dev(shape=(h, w), dtype=np.uint8, pitched=True)
    lib().baoCudaCensusTransform(c1.ptr, c2.ptr, a.ptr, b.ptr, w, h, _sz(a.pitch), _sz(c1.pitch))
    check_launcher("baoCudaCensusTransform")
    return c1.get(), c2.get()


def prepare(raw1, raw2, dims):
    """baoCudaPatchMatchMultiscalePrepare on raw RGBA planes; dims = [(h,w)] per level.  Returns (imgs1, imgs2, cens1, cens2)."""
    n = len(dims)
    h, w = dims[0]
    r1, r2 = Dev(raw1, pitched=True), Dev(raw2, pitched=True)
    p1 = [Dev(shape=d, dtype=uchar4, pitched=True) for d in dims]
    p2 = [Dev(shape=d, dtype=uchar4, pitched=True) for d in dims]
    t1 = [Dev(shape=d, dtype=uchar4, pitched=True) for d in dims]
    t2 = [Dev(shape=d, dtype=uchar4, pitched=True) for d in dims]
    c1 = [Dev(shape=d, dtype=np.uint8, pitched=True) for d in dims]
    c2 = [Dev(shape=d, dtype=np.uint8, pitched=True) for d in dims]

    def tab(bufs):
        return (C.c_void_p * n)(*[b.ptr for b in bufs])

    arrH = (C.c_int * n)(*[d[0] for d in dims])
    arrW = (C.c_int * n)(*[d[1] for d in dims])
    p4 = (C.c_size_t * n)(*[b.pitch for b in p1])
    pc = (C.c_size_t * n)(*[b.pitch for b in c1])
    lib().baoCudaPatchMatchMultiscalePrepare(tab(p1), tab(p2), tab(c1), tab(c2), tab(t1), tab(t2), arrH, arrW, p4, pc, n,
                                             r1.ptr, r2.ptr, h, w)
    check_launcher("baoCudaPatchMatchMultiscalePrepare")
    return [b.get() for b in p1], [b.get() for b in p2], [b.get() for b in c1], [b.get() for b in c2]


class PlaneSet:
    """Device copies of (img1, img2, census1, census2) of one level."""

    def __init__(self, img1, img2, c1, c2):
        self.h, self.w = img1.shape
        self.i1, self.i2 = Dev(img1, pitched=True), Dev(img2, pitched=True)
        self.c1, self.c2 = Dev(c1, pitched=True), Dev(c2, pitched=True)

    def args(self):
        return (self.i1.ptr, self.i2.ptr, self.c1.ptr, self.c2.ptr)


class PmRng:
    def __init__(self, w, h, params=None):
        self.p = C.c_void_p()
        self.w, self.h = w, h
        check(lib().eppm_pm_rng_create(C.byref(self.p), w, h, C.byref(params) if params is not None else None), "rng_create")

    def block_states(self):
        nb = ((self.w + 15) // 16) * ((self.h + 15) // 16)
        out = np.empty((nb, 6), np.uint32)
        check(lib().eppm_pm_rng_block_states(self.p, out.ctypes.data_as(C.c_void_p), _sz(out.size)), "rng_block_states")
        return out

    def __del__(self):
        try:
            if self.p:
                lib().eppm_pm_rng_destroy(self.p)
        except Exception:
            pass


def pm_gen_rand_field(rng):
    nnf = Dev(shape=(rng.h, rng.w), dtype=short2)
    check(lib().eppm_memset_device(nnf.ptr, 0, _sz(nnf.pitch * nnf.h)), "memset")
    check(lib().eppm_pm_gen_rand_field(rng.p, nnf.ptr, rng.w, rng.h, _sz(nnf.pitch)), "pm_gen_rand_field")
    return nnf.get()


def pm_cost_field(nnf, P):
    n, c = Dev(nnf), Dev(shape=(P.h, P.w), dtype=np.float32)
    check(lib().eppm_pm_cost_field(c.ptr, n.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch), _sz(P.c1.pitch)),
          "pm_cost_field")
    return c.get()


def pm_seg_propagate(cost, nnf, P, direction):
    n, c = Dev(nnf), Dev(cost)
    check(lib().eppm_pm_seg_propagate(c.ptr, n.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch), _sz(P.c1.pitch),
                                      direction), "pm_seg_propagate")
    return c.get(), n.get()


def pm_jump_propagate(cost, nnf, P):
    n, c = Dev(nnf), Dev(cost)
    check(lib().eppm_pm_jump_propagate(c.ptr, n.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch), _sz(P.c1.pitch)),
          "pm_jump_propagate")
    return c.get(), n.get()


def pm_parallel_propagate(cost, nnf, P):
    """One launch of the 4-neighbour propagation (baoParallelPropagate, bao_pmflow_kernel.cu:720-795)."""
    n, c = Dev(nnf), Dev(cost)
    check(lib().eppm_pm_parallel_propagate(c.ptr, n.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch),
                                           _sz(P.c1.pitch)), "pm_parallel_propagate")
    return c.get(), n.get()


def pm_random_search(rng, cost, nnf, P):
    n, c = Dev(nnf), Dev(cost)
    check(lib().eppm_pm_random_search(rng.p, c.ptr, n.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch),
                                      _sz(P.c1.pitch)), "pm_random_search")
    return c.get(), n.get()


def patchmatch(P):
    """baoCudaPatchMatch -> (nnf, cost)"""
    n, c = Dev(shape=(P.h, P.w), dtype=short2), Dev(shape=(P.h, P.w), dtype=np.float32)
    lib().baoCudaPatchMatch(n.ptr, c.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(c.pitch), _sz(n.pitch), _sz(P.c1.pitch))
    check_launcher("baoCudaPatchMatch")
    return n.get(), c.get()


def left_right_check(nnf1, cost1, nnf2, cost2):
    h, w = nnf1.shape
    a, b, c, d = Dev(nnf1), Dev(cost1), Dev(nnf2), Dev(cost2)
    lib().baoCudaLeftRightCheck(a.ptr, b.ptr, c.ptr, d.ptr, w, h, _sz(b.pitch), _sz(a.pitch))
    check_launcher("baoCudaLeftRightCheck")
    return a.get(), b.get(), c.get(), d.get()


def outlier_removal(nnf, cost):
    h, w = nnf.shape
    a, b = Dev(nnf), Dev(cost)
    lib().baoCudaOutlierRemoval(a.ptr, b.ptr, w, h, _sz(b.pitch), _sz(a.pitch))
    check_launcher("baoCudaOutlierRemoval")
    return a.get(), b.get()


def weighted_median(nnf, img, num_iter=20, only_occlusion=True):
    h, w = nnf.shape
    a, i = Dev(nnf), Dev(img, pitched=True)
    lib().baoCudaWeightedMedianFilter(a.ptr, None, i.ptr, w, h, _sz(i.pitch), _sz(w * 4), _sz(a.pitch), num_iter, C.c_bool(only_occlusion))
    check_launcher("baoCudaWeightedMedianFilter")
    return a.get()


def fill_holes(nnf, img):
    h, w = nnf.shape
    a, i = Dev(nnf), Dev(img, pitched=True)
    lib().baoCudaFillHole(a.ptr, None, i.ptr, w, h, _sz(i.pitch), _sz(w * 4), _sz(a.pitch))
    check_launcher("baoCudaFillHole")
    return a.get()


def nnf2flow(nnf):
    h, w = nnf.shape
    a, f = Dev(nnf), Dev(shape=(h, w), dtype=float2)
    lib().baoCudaNNF2Flow(f.ptr, a.ptr, w, h, _sz(a.pitch), _sz(f.pitch))
    check_launcher("baoCudaNNF2Flow")
    return f.get()


def resize_flow(flow, out_h, out_w, ratio=2.0):
    h, w = flow.shape
    a, b = Dev(flow), Dev(shape=(out_h, out_w), dtype=float2)
    check(lib().eppm_resize_flow(b.ptr, out_h, out_w, a.ptr, h, w, C.c_float(ratio)), "resize_flow")
    return b.get()


def c2f_refine(flow, P):
    """baoCudaBLFCostFilterRefine"""
    f = Dev(flow)
    lib().baoCudaBLFCostFilterRefine(f.ptr, *P.args(), P.w, P.h, _sz(P.i1.pitch), _sz(P.c1.pitch))
    check_launcher("baoCudaBLFCostFilterRefine")
    return f.get()


def blf_c2f(flow_coarse, P_fine, coarse_dims):
    """baoCudaBLF_C2F from level l+1 (flow_coarse) to level l (P_fine): upsample x2, x2.0, candidate refine."""
    ch, cw = coarse_dims
    fine = Dev(shape=(P_fine.h, P_fine.w), dtype=float2)
    coarse = Dev(flow_coarse)
    tab = lambda a, b: (C.c_void_p * 2)(a, b)  # noqa: E731
    arrH = (C.c_int * 2)(P_fine.h, ch)
    arrW = (C.c_int * 2)(P_fine.w, cw)
    p4 = (C.c_size_t * 2)(P_fine.i1.pitch, 0)
    pc = (C.c_size_t * 2)(P_fine.c1.pitch, 0)
    lib().baoCudaBLF_C2F(tab(fine.ptr, coarse.ptr), tab(P_fine.i1.ptr, None), tab(P_fine.i2.ptr, None), tab(P_fine.c1.ptr, None),
                         tab(P_fine.c2.ptr, None), None, None, arrH, arrW, p4, pc, 0)
    check_launcher("baoCudaBLF_C2F")
    return fine.get()


def flow_smoothing(flow, img):
    h, w = flow.shape
    f, i = Dev(flow), Dev(img, pitched=True)
    lib().baoCudaFlowSmoothing(f.ptr, i.ptr, w, h, _sz(i.pitch), _sz(f.pitch))
    check_launcher("baoCudaFlowSmoothing")
    return f.get()


def flow_to_color(flow, max_disp_x=20.0, max_disp_y=20.0):
    """bao_cuda_convert_flow_to_colorshow (float2 form, basic/bao_basic_cuda.cuh:839-845): (h,w) float2 -> (h,w) uchar4 {R,G,B,0}."""
    h, w = flow.shape
    f, c = Dev(flow), Dev(shape=(h, w), dtype=uchar4)
    check(lib().eppm_flow_to_color(c.ptr, f.ptr, h, w, C.c_float(max_disp_x), C.c_float(max_disp_y)), "eppm_flow_to_color")
    return c.get()
