"""ctypes binding of the C ABI declared in include/eppm.h.  Fails loudly if the library is missing."""
import ctypes as C
import os

from .build import lib_path


class EppmError(RuntimeError):
    pass


class CParams(C.Structure):
    _fields_ = [("patch_r", C.c_int), ("num_iter", C.c_int), ("search_range", C.c_int), ("num_guess", C.c_int),
                ("seg_len", C.c_int), ("wmf_iters", C.c_int), ("seed", C.c_ulonglong), ("propagation", C.c_int),
                ("levels", C.c_int)]


_lib = None

# every symbol include/eppm.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "eppm_default_params", "eppm_create", "eppm_create_batch", "eppm_batch_size", "eppm_batch_set_images", "eppm_batch_set_images_device",
    "eppm_batch_compute", "eppm_batch_compute_device", "eppm_batch_compute_end", "eppm_batch_get_plane", "eppm_destroy", "eppm_set_stream", "eppm_set_images", "eppm_set_images_device",
    "eppm_compute", "eppm_compute_begin", "eppm_compute_end", "eppm_compute_device", "eppm_synchronize", "eppm_num_levels", "eppm_level_dims", "eppm_get_plane",
    "eppm_stage_times", "eppm_clear_stage_times", "eppm_enable_stage_timing", "eppm_last_error", "eppm_version",
    "eppm_device_count", "eppm_set_device", "eppm_malloc_device", "eppm_malloc_pitched", "eppm_free_device",
    "eppm_memcpy_h2d", "eppm_memcpy_d2h", "eppm_memcpy2d_h2d", "eppm_memcpy2d_d2h", "eppm_memset_device",
    "eppm_device_synchronize", "eppm_device_mem_info", "eppm_device_pci_bus_id", "eppm_bind_thread_to_device", "eppm_release_cached_memory", "eppm_set_launcher_stream", "eppm_set_launcher_params", "eppm_launcher_status",
    "baoCudaPatchMatchMultiscalePrepare", "baoCudaCensusTransform", "baoCudaPatchMatch", "baoCudaLeftRightCheck",
    "baoCudaOutlierRemoval", "baoCudaWeightedMedianFilter", "baoCudaFillHole", "baoCudaNNF2Flow", "baoCudaBLF_C2F",
    "baoCudaBLFCostFilterRefine", "baoCudaFlowSmoothing", "eppm_flow_to_color", "eppm_compute_color",
    "eppm_pm_rng_create", "eppm_pm_rng_reset", "eppm_pm_rng_destroy", "eppm_pm_rng_block_states", "eppm_pm_gen_rand_field",
    "eppm_pm_cost_field", "eppm_pm_seg_propagate", "eppm_pm_jump_propagate", "eppm_pm_parallel_propagate", "eppm_pm_random_search", "eppm_gauss_filter_rgba", "eppm_resize_rgba",
    "eppm_resize_flow",
    "eppm_host_register", "eppm_host_unregister", "eppm_host_is_registered", "eppm_host_alloc", "eppm_host_free",
    "eppm_compute_begin_into", "eppm_batch_compute_begin_into",
    "eppm_load_ppm", "eppm_ppm_size", "eppm_save_flo", "eppm_load_flo", "eppm_flo_size", "eppm_flow_error",
    "eppm_flow_error_border", "eppm_flow_error_percentage", "eppm_flow_cutoff", "eppm_flow_to_color_host",
]


# what include/eppm_test.h adds, exported by libeppm_hip_test.so only (the parity tests' switches and arithmetic probes)
TEST_SYMBOLS = ["eppm_test_set_option", "eppm_probe_c2f_window", "eppm_probe_fast_exp", "eppm_probe_div_const", "eppm_probe_delta_table"]

_variant = None


def select_library(variant):
    """Which build this PROCESS loads: "" the product library (default), "test" libeppm_hip_test.so (the same objects plus the test
    hooks of include/eppm_test.h: tests/conftest.py selects it for the pytest process; child processes -- bench.py, the CLI, smoke() --
    are not affected), "tol" the tolerance library (libeppm_hip_tol.so, not bit-identical).  Must be called before the first lib()."""
    global _variant
    if _lib is not None and variant != _variant:
        raise EppmError("select_library: a library is loaded already")
    _variant = variant


def lib():
    """Load libeppm_hip.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        path = lib_path(_variant)
        if not os.path.exists(path):
            raise EppmError(f"{path} is missing: run eppm_amd.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(path)
        L.eppm_last_error.restype = C.c_char_p
        L.eppm_version.restype = C.c_char_p
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        raise EppmError(f"{what}: status {status}: {lib().eppm_last_error().decode()}")


def check_launcher(what=""):
    check(lib().eppm_launcher_status(), what)
