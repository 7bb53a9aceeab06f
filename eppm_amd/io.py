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


def flow_error(u, v, gu, gv, border=0):
    """(EPE, AAE) with the reference's validity rule (bao_calc_flow_error); border: pixels left out on every side."""
    arrs = [np.ascontiguousarray(a, np.float32) for a in (u, v, gu, gv)]
    h, w = arrs[0].shape
    epe, aae = C.c_float(), C.c_float()
    check(lib().eppm_flow_error_border(*[a.ctypes.data_as(C.c_void_p) for a in arrs], h, w, int(border), C.byref(epe), C.byref(aae)), "eppm_flow_error_border")
    return epe.value, aae.value


def flow_error_percentage(u, v, gu, gv, error_thresh, want_map=False):
    """Fraction of the pixels with known ground truth whose end-point error exceeds error_thresh (bao_calc_flow_error_percentage);
    with want_map also the uint8 map (255 where it does)."""
    arrs = [np.ascontiguousarray(a, np.float32) for a in (u, v, gu, gv)]
    h, w = arrs[0].shape
    emap = np.empty((h, w), np.uint8) if want_map else None
    frac = C.c_float()
    check(lib().eppm_flow_error_percentage(*[a.ctypes.data_as(C.c_void_p) for a in arrs], h, w, int(error_thresh),
                                           emap.ctypes.data_as(C.c_void_p) if want_map else None, C.byref(frac)), "eppm_flow_error_percentage")
    return (frac.value, emap) if want_map else frac.value


def flow_cutoff(u, v, cutoff, cut_invalid=False):
    """Both components clamped to [-|cutoff|, |cutoff|]; unknown vectors pass through unless cut_invalid (bao_flow_cutoff)."""
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    uo, vo = np.empty_like(u), np.empty_like(v)
    h, w = u.shape
    check(lib().eppm_flow_cutoff(uo.ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p),
                                 v.ctypes.data_as(C.c_void_p), h, w, int(cutoff), int(bool(cut_invalid))), "eppm_flow_cutoff")
    return uo, vo


def flow_to_color(u, v):
    """(h, w, 3) uint8 R,G,B colour coding scaled by the field's largest known radius (bao_convert_flow_to_colorshow, host)."""
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    h, w = u.shape
    rgb = np.empty((h, w, 3), np.uint8)
    check(lib().eppm_flow_to_color_host(rgb.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), h, w),
          "eppm_flow_to_color_host")
    return rgb
