"""Build libeppm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def lib_path():
    return os.path.join(_HERE, "lib", "libeppm_hip.so")


def _stale():
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("eppm.h", "bao_flow_patchmatch_multiscale_cuda.h")]
    srcs.append(os.path.join(_HERE, "..", "tools", "runeppm.cpp"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def build(force=False, verbose=False):
    """Compile every HIP kernel + the C ABI into eppm_amd/lib/libeppm_hip.so (and the runeppm CLI)."""
    if not force and not _stale():
        return lib_path()
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        if os.path.exists(lib_path()):
            return lib_path()
        raise RuntimeError("hipcc not found and no prebuilt libeppm_hip.so")
    cmd = ["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return lib_path()
