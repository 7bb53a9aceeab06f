"""Build libeppm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def lib_path(variant=None):
    """The product library; variant="test": the parity tests' library (libeppm_hip_test.so: the same objects + include/eppm_test.h);
    variant="approx": the opt-in approx-exp build.  variant=None reads EPPM_HIP_VARIANT from the environment."""
    variant = variant if variant is not None else os.environ.get("EPPM_HIP_VARIANT", "")
    if variant not in ("", "exact", "approx", "test"):
        raise ValueError(f"unknown library variant {variant!r}")
    name = {"approx": "libeppm_hip_approx.so", "test": "libeppm_hip_test.so"}.get(variant, "libeppm_hip.so")
    return os.path.join(_HERE, "lib", name)


def _stale():
    outs = [lib_path(""), lib_path("test")]
    if not all(os.path.exists(o) for o in outs):
        return True
    t = min(os.path.getmtime(o) for o in outs)
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("eppm.h", "eppm_test.h", "bao_flow_patchmatch_multiscale_cuda.h", "bao_basic_cuda.h")]
    srcs.append(os.path.join(_HERE, "..", "tools", "runeppm.cpp"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def build(force=False, verbose=False, approx=False):
    """Compile every HIP kernel + the C ABI into eppm_amd/lib/libeppm_hip.so, the parity tests' libeppm_hip_test.so and the runeppm CLI; approx=True also builds the opt-in
    libeppm_hip_approx.so (`make approx`: v_exp_f32 arithmetic, not bit-identical, never part of the default build)."""
    if approx:
        subprocess.check_call(["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1)), "approx"], stdout=None if verbose else subprocess.DEVNULL)
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
