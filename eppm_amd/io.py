"""PPM / .flo I/O and EPE through the C ABI (reference: bao_basic.cpp:137-218, flowIO.cpp:48-163,
bao_flow_tools.cpp:64-111)."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def ppm_size(path):
    h, w = C.c_int(), C.c_int()
    check(lib().eppm_ppm_size(path.encode(), C.byref(h), C.byref(w)), f"eppm_ppm_size({path})")
    return h.value, w.value


def load_ppm(path):
    h, w = ppm_size(path)
    img = np.zeros((h, w, 3), np.uint8)
    nc = C.c_int()
    check(lib().eppm_load_ppm(path.encode(), img.ctypes.data_as(C.c_void_p), h, w, C.byref(nc)), f"eppm_load_ppm({path})")
    return img if nc.value == 3 else img.reshape(-1)[:h * w].reshape(h, w)


def save_flo(path, u, v):
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    h, w = u.shape
    check(lib().eppm_save_flo(path.encode(), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), h, w), "eppm_save_flo")


def load_flo(path):
    h, w = C.c_int(), C.c_int()
    check(lib().eppm_flo_size(path.encode(), C.byref(h), C.byref(w)), f"eppm_flo_size({path})")
    u = np.empty((h.value, w.value), np.float32)
    v = np.empty((h.value, w.value), np.float32)
    check(lib().eppm_load_flo(path.encode(), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), h.value, w.value), "eppm_load_flo")
    return u, v


def flow_error(u, v, gu, gv):
    """(EPE, AAE) with the reference's validity rule."""
    arrs = [np.ascontiguousarray(a, np.float32) for a in (u, v, gu, gv)]
    h, w = arrs[0].shape
    epe, aae = C.c_float(), C.c_float()
    check(lib().eppm_flow_error(*[a.ctypes.data_as(C.c_void_p) for a in arrs], h, w, C.byref(epe), C.byref(aae)), "eppm_flow_error")
    return epe.value, aae.value
