"""Host-side mirror of ``class bao_flow_patchmatch_multiscale_cuda`` over the C ABI."""
import ctypes as C

import numpy as np

from ._lib import CParams, EppmError, check, lib

uchar4 = np.dtype([("x", "u1"), ("y", "u1"), ("z", "u1"), ("w", "u1")])
short2 = np.dtype([("x", "i2"), ("y", "i2")])
float2 = np.dtype([("x", "f4"), ("y", "f4")])

_PLANE_DTYPES = {"img1": uchar4, "img2": uchar4, "census1": np.uint8, "census2": np.uint8, "nnf1": short2,
                 "nnf2": short2, "cost1": np.float32, "cost2": np.float32, "flow": float2}


def Params(**kw):
    """Tunables with the defaults of defs.h:31-76 (patch_r, num_iter, search_range, num_guess, seg_len, wmf_iters, seed, propagation: 0 segmented sweeps / 1 jump flood / 2 4-neighbour, levels: pyramid depth)."""
    p = CParams()
    check(lib().eppm_default_params(C.byref(p)), "eppm_default_params")
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown parameter {k}")
        setattr(p, k, v)
    return p


def host_register(arr):
    """Pin a numpy array's memory for DMA (eppm_host_register): set_data reads registered images, and compute_flow(out=...)
    writes registered planes, over PCIe directly -- no staging copy.  Keep the array alive until host_unregister."""
    check(lib().eppm_host_register(C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)), "eppm_host_register")
    return arr


def host_unregister(arr):
    check(lib().eppm_host_unregister(C.c_void_p(arr.ctypes.data)), "eppm_host_unregister")


def pinned_empty(shape, dtype=np.uint8):
    """A numpy array in pinned host memory from eppm_host_alloc (counts as registered); freed with the array."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    check(lib().eppm_host_alloc(C.byref(p), C.c_size_t(max(n, 1))), "eppm_host_alloc")
    addr = p.value

    class _Owner:
        def __del__(self):
            try:
                lib().eppm_host_free(C.c_void_p(addr))
            except Exception:
                pass
    buf = (C.c_char * max(n, 1)).from_address(addr)
    buf._owner = _Owner()
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class EPPM:
    """``init`` / ``set_data`` / ``compute_flow`` as in bao_flow_patchmatch_multiscale_cuda.h:36-44.

    Images are (h, w, 3) uint8 arrays in R,G,B order (the reference's ``unsigned char***``);
    flows are (h, w) float32 arrays (``float**``).
    """

    def __init__(self, device=0, params=None):
        self._ctx = C.c_void_p()
        self._device = device
        self._params = params
        self._pending_out = None
        self.h = self.w = 0

    # -- reference interface ---------------------------------------------------------------
    def init(self, *args):
        """init(h, w)  or  init(img1, img2, h, w)   (driver .cpp:106-157)"""
        if len(args) == 4:
            img1, img2, h, w = args
            self.init(h, w)
            self.set_data(img1, img2)
            return
        h, w = args
        self.close()
        ctx = C.c_void_p()
        check(lib().eppm_create(C.byref(ctx), int(h), int(w), int(self._device),
                                C.byref(self._params) if self._params is not None else None), "eppm_create")
        self._ctx, self.h, self.w = ctx, int(h), int(w)

    def set_data(self, img1, img2):
        """RGB->RGBA, H2D, prefilter, pyramid, census (driver .cpp:159-168).  Returns True like the reference."""
        self._need()
        a = np.ascontiguousarray(img1, np.uint8)
        b = np.ascontiguousarray(img2, np.uint8)
        if a.shape != (self.h, self.w, 3) or b.shape != (self.h, self.w, 3):
            raise EppmError(f"set_data: images must be ({self.h},{self.w},3) uint8")
        check(lib().eppm_set_images(self._ctx, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                    C.c_size_t(self.w * 3)), "eppm_set_images")
        return True

    def _out(self, out):
        if out is None:
            return np.empty((self.h, self.w), np.float32), np.empty((self.h, self.w), np.float32)
        u, v = out
        for a in (u, v):
            if a.shape != (self.h, self.w) or a.dtype != np.float32 or not a.flags.c_contiguous:
                raise EppmError(f"out planes must be C-contiguous ({self.h},{self.w}) float32")
        return u, v

    def compute_flow(self, out=None):
        """Returns (disp1_x, disp1_y) (driver .cpp:217-306); out=(u, v): write into these planes (registered planes are written
        by DMA directly, see host_register)."""
        self._need()
        u, v = self._out(out)
        check(lib().eppm_compute(self._ctx, u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "eppm_compute")
        return u, v

    def compute_flow_color(self, max_disp=(20.0, 20.0)):
        """The optional color_flow output of compute_flow (driver .cpp:308-314): (h, w, 3) uint8 R,G,B of the last flow."""
        self._need()
        rgb = np.empty((self.h, self.w, 3), np.uint8)
        check(lib().eppm_compute_color(self._ctx, rgb.ctypes.data_as(C.c_void_p), C.c_size_t(self.w * 3),
                                       C.c_float(max_disp[0]), C.c_float(max_disp[1])), "eppm_compute_color")
        return rgb

    def compute_flow_begin(self, out=None):
        """Enqueue compute_flow and its device-to-host copy, return at once (eppm_compute_begin; with out=(u, v):
        eppm_compute_begin_into, which lets registered planes receive the copy directly)."""
        self._need()
        if out is None:
            check(lib().eppm_compute_begin(self._ctx), "eppm_compute_begin")
        else:
            u, v = self._out(out)
            self._pending_out = (u, v)      # the copy engine writes these planes until compute_flow_end: keep them alive
            check(lib().eppm_compute_begin_into(self._ctx, u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "eppm_compute_begin_into")

    def compute_flow_end(self, out=None):
        """Wait for compute_flow_begin; returns (disp1_x, disp1_y)."""
        self._need()
        u, v = self._out(out)
        check(lib().eppm_compute_end(self._ctx, u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)), "eppm_compute_end")
        self._pending_out = None
        return u, v

    # -- device-resident variants (no PCIe in the timed region) ----------------------------
    def set_data_device(self, d_rgba1, d_rgba2, pitch):
        self._need()
        check(lib().eppm_set_images_device(self._ctx, C.c_void_p(d_rgba1), C.c_void_p(d_rgba2), C.c_size_t(pitch)),
              "eppm_set_images_device")

    def compute_flow_device(self, d_flow=None):
        self._need()
        check(lib().eppm_compute_device(self._ctx, C.c_void_p(d_flow) if d_flow else None), "eppm_compute_device")

    def synchronize(self):
        self._need()
        check(lib().eppm_synchronize(self._ctx), "eppm_synchronize")

    def set_stream(self, hip_stream):
        self._need()
        check(lib().eppm_set_stream(self._ctx, C.c_void_p(hip_stream)), "eppm_set_stream")

    # -- introspection ---------------------------------------------------------------------
    def level_dims(self):
        self._need()
        out = []
        for l in range(lib().eppm_num_levels(self._ctx)):
            h, w = C.c_int(), C.c_int()
            check(lib().eppm_level_dims(self._ctx, l, C.byref(h), C.byref(w)), "eppm_level_dims")
            out.append((h.value, w.value))
        return out

    def plane(self, name, level):
        self._need()
        dims = self.level_dims()
        if name not in _PLANE_DTYPES or not 0 <= level < len(dims):
            raise EppmError(f"plane({name!r}, {level}): unknown plane or level")
        h, w = dims[level]
        a = np.empty((h, w), _PLANE_DTYPES[name])
        check(lib().eppm_get_plane(self._ctx, name.encode(), level, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)),
              "eppm_get_plane")
        return a

    def enable_stage_timing(self, on=True):
        self._need()
        check(lib().eppm_enable_stage_timing(self._ctx, int(on)), "eppm_enable_stage_timing")

    def stage_times(self, clear=True):
        """[(stage name, ms)] for every set_data / compute_flow since the last clear (timing enabled)."""
        self._need()
        cap = 1 << 16
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = lib().eppm_stage_times(self._ctx, names, ms, cap)
        out = [(names[i].decode(), float(ms[i])) for i in range(n)]
        if clear:
            check(lib().eppm_clear_stage_times(self._ctx), "eppm_clear_stage_times")
        return out

    def close(self):
        if self._ctx:
            lib().eppm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def _need(self):
        if not self._ctx:
            raise EppmError("EPPM: init(h, w) has not been called")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EPPMBatch:
    """A batch of independent pairs of one size on one GPU (eppm_create_batch): every kernel launch of the path covers all
    active pairs.  ``set_data(pairs)`` / ``compute_flow()`` mirror the single-pair class; each pair's flow is bit-identical
    to the single-pair result."""

    def __init__(self, h, w, npairs, device=0, params=None):
        self._ctx = C.c_void_p()
        check(lib().eppm_create_batch(C.byref(self._ctx), int(h), int(w), int(device),
                                      C.byref(params) if params is not None else None, int(npairs)), "eppm_create_batch")
        self.h, self.w, self.npairs, self.n = int(h), int(w), int(npairs), 0
        self._pending_out = None

    @staticmethod
    def _ptrs(arrs):
        return (C.c_void_p * len(arrs))(*[a.ctypes.data if hasattr(a, "ctypes") else a for a in arrs])

    def set_data(self, pairs):
        """pairs: up to npairs (img1, img2) tuples of (h, w, 3) uint8 arrays."""
        a = [np.ascontiguousarray(p[0], np.uint8) for p in pairs]
        b = [np.ascontiguousarray(p[1], np.uint8) for p in pairs]
        for x in a + b:
            if x.shape != (self.h, self.w, 3):
                raise EppmError(f"set_data: images must be ({self.h},{self.w},3) uint8")
        check(lib().eppm_batch_set_images(self._ctx, len(pairs), self._ptrs(a), self._ptrs(b), C.c_size_t(self.w * 3)), "eppm_batch_set_images")
        self.n = len(pairs)

    def set_data_device(self, d1, d2, pitch):
        """d1, d2: lists of device addresses of RGBA planes."""
        check(lib().eppm_batch_set_images_device(self._ctx, len(d1), self._ptrs(list(d1)), self._ptrs(list(d2)), C.c_size_t(pitch)),
              "eppm_batch_set_images_device")
        self.n = len(d1)

    def _outs(self, out=None):
        if out is not None:
            u, v = [o[0] for o in out], [o[1] for o in out]
            for a in u + v:
                if a.shape != (self.h, self.w) or a.dtype != np.float32 or not a.flags.c_contiguous:
                    raise EppmError(f"out planes must be C-contiguous ({self.h},{self.w}) float32")
            if len(u) < self.n:
                raise EppmError("out: one (u, v) per active pair")
            return u[:self.n], v[:self.n]
        u = [np.empty((self.h, self.w), np.float32) for _ in range(self.n)]
        v = [np.empty((self.h, self.w), np.float32) for _ in range(self.n)]
        return u, v

    def compute_flow(self, out=None):
        """[(u, v)] for the active pairs; out: list of (u, v) planes to write into (registered planes receive the DMA directly)."""
        u, v = self._outs(out)
        check(lib().eppm_batch_compute(self._ctx, self._ptrs(u), self._ptrs(v)), "eppm_batch_compute")
        return list(zip(u, v))

    def compute_flow_device(self, d_flows=None):
        check(lib().eppm_batch_compute_device(self._ctx, self._ptrs(list(d_flows)) if d_flows is not None else None), "eppm_batch_compute_device")

    def compute_flow_begin(self, out=None):
        if out is None:
            check(lib().eppm_compute_begin(self._ctx), "eppm_compute_begin")
        else:
            u, v = self._outs(out)
            self._pending_out = (u, v)      # the copy engine writes these planes until compute_flow_end: keep them alive
            check(lib().eppm_batch_compute_begin_into(self._ctx, self._ptrs(u), self._ptrs(v)), "eppm_batch_compute_begin_into")

    def compute_flow_end(self, out=None):
        u, v = self._outs(out)
        check(lib().eppm_batch_compute_end(self._ctx, self._ptrs(u), self._ptrs(v)), "eppm_batch_compute_end")
        self._pending_out = None
        return list(zip(u, v))

    def synchronize(self):
        check(lib().eppm_synchronize(self._ctx), "eppm_synchronize")

    def plane(self, pair, name, level):
        dims = []
        for l in range(lib().eppm_num_levels(self._ctx)):
            hh, ww = C.c_int(), C.c_int()
            check(lib().eppm_level_dims(self._ctx, l, C.byref(hh), C.byref(ww)), "eppm_level_dims")
            dims.append((hh.value, ww.value))
        hh, ww = dims[level]
        a = np.empty((hh, ww), _PLANE_DTYPES[name])
        check(lib().eppm_batch_get_plane(self._ctx, int(pair), name.encode(), level, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)),
              "eppm_batch_get_plane")
        return a

    def enable_stage_timing(self, on=True):
        check(lib().eppm_enable_stage_timing(self._ctx, int(on)), "eppm_enable_stage_timing")

    def stage_times(self, clear=True):
        cap = 1 << 16
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = lib().eppm_stage_times(self._ctx, names, ms, cap)
        out = [(names[i].decode(), float(ms[i])) for i in range(n)]
        if clear:
            check(lib().eppm_clear_stage_times(self._ctx), "eppm_clear_stage_times")
        return out

    def close(self):
        if self._ctx:
            lib().eppm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
