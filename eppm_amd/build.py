"""Build libeppm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def lib_path(variant=None):
    """The product library (bit-identical to the oracle); variant="test": the parity tests' library (libeppm_hip_test.so: the same objects +
    include/eppm_test.h); variant="tol": the tolerance library (libeppm_hip_tol.so: integer-domain tables in the patch term, NOT
    bit-identical, inside 1e-3 px EPE on the bundled pair; never the default).  variant=None reads EPPM_HIP_VARIANT from the environment."""
    variant = variant if variant is not None else os.environ.get("EPPM_HIP_VARIANT", "")
    if variant not in ("", "exact", "tol", "test"):
        raise ValueError(f"unknown library variant {variant!r}")
    name = {"tol": "libeppm_hip_tol.so", "test": "libeppm_hip_test.so"}.get(variant, "libeppm_hip.so")
    return os.path.join(_HERE, "lib", name)


def _stale():
    outs = [lib_path(""), lib_path("test"), lib_path("tol")]
    if not all(os.path.exists(o) for o in outs):
        return True
    t = min(os.path.getmtime(o) for o in outs)
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("eppm.h", "eppm_test.h", "bao_flow_patchmatch_multiscale_cuda.h", "bao_basic_cuda.h")]
    srcs.append(os.path.join(_HERE, "..", "tools", "runeppm.cpp"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def build(force=False, verbose=False):
    """Compile every HIP kernel + the C ABI into eppm_amd/lib/libeppm_hip.so, the parity tests' libeppm_hip_test.so, the tolerance library
    libeppm_hip_tol.so (the same sources with -DEPPM_TOL; never loaded unless asked for) and the runeppm CLI."""
    if not force and not _stale():
        return lib_path("")
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        if os.path.exists(lib_path("")):
            return lib_path("")
        raise RuntimeError("hipcc not found and no prebuilt libeppm_hip.so")
    cmd = ["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return lib_path("")
