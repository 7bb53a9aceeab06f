"""CPU tests of the C-ABI library: it loads, exports every symbol include/eppm.h declares, its argument
checks work without a GPU, and its file I/O equals the reference's own code (oracle/_ref) byte for byte."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
import eppm_amd
from eppm_amd import _lib
from oracle import oracle as O


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b((?:eppm|baoCuda)\w*)\s*\(", hdr))


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_library_exports_every_declared_symbol():
    declared = _declared("eppm.h")
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    P = C.CDLL(eppm_amd.lib_path(""))          # the PRODUCT library (this process otherwise works on the test library, conftest.py)
    for s in declared:
        getattr(P, s)          # raises AttributeError if not exported
    P.eppm_version.restype = C.c_char_p
    assert b"tolerance" not in P.eppm_version()
    # the tolerance library (same sources, -DEPPM_TOL; never loaded by default) exports the same ABI and names itself
    A = C.CDLL(eppm_amd.lib_path("tol"))
    for s in declared:
        getattr(A, s)
    A.eppm_version.restype = C.c_char_p
    assert b"tolerance arithmetic" in A.eppm_version() and b"not bit-identical" in A.eppm_version()
    assert not set(_exported(eppm_amd.lib_path("tol"))) & set(_lib.TEST_SYMBOLS)          # no test hooks in it either


def test_test_hooks_live_in_the_test_library_only():
    """include/eppm_test.h = the switches and probes of the parity tests.  The product library exports none of them (nothing matching
    eppm_test* / eppm_probe*), the test library exports them on top of everything include/eppm.h declares, and the two are linked from the
    same objects except eppm_api.o (compiled with / without -DEPPM_TEST_HOOKS) and the probe kernel."""
    hooks = _declared("eppm_test.h")
    assert hooks == set(_lib.TEST_SYMBOLS) and not (hooks & _declared("eppm.h"))
    prod, test = _exported(eppm_amd.lib_path("")), _exported(eppm_amd.lib_path("test"))
    assert not [s for s in prod if s.startswith("eppm_test") or s.startswith("eppm_probe")]
    assert hooks <= test and _declared("eppm.h") <= test and _declared("eppm.h") <= prod
    assert {s for s in test - prod if not s.startswith("_")} == hooks          # nothing else differs in the exported C ABI
    mk = open(os.path.join(ROOT, "eppm_amd", "csrc", "Makefile")).read()
    assert "HOOKED = rng_tables context launchers_ref_abi" in mk
    assert "OBJS_T = $(filter-out $(HOOKED:%=$(OBJ)/%.o),$(OBJS)) $(HOOKED:%=$(OBJ)/%_test.o) $(OBJ)/test_hooks.o $(OBJ)/k_probe.o" in mk
    csrc = os.path.join(ROOT, "eppm_amd", "csrc")            # the switches are read where the Makefile says, and nowhere else
    readers = {f[:-4] for f in os.listdir(csrc) if f.endswith(".cpp") and f != "test_hooks.cpp" and re.search(r"\bopt_(rand_table|sweep_spec|no_split)\(\)", open(os.path.join(csrc, f)).read())}
    assert readers == {"rng_tables", "context", "launchers_ref_abi"}, readers
    assert eppm_amd.lib()._name == eppm_amd.lib_path("test")                   # what the pytest process itself computes with


def test_host_registry_under_thread_sanitizer(tmp_path):
    """The registry of caller memory pinned for DMA (eppm_amd/csrc/host_registry.h: refcounts, aliases, `closing`, bounded waits) is plain
    host C++ with the two runtime calls injected, so it runs here under ThreadSanitizer: 8 threads x 100 000 random register / inner-range
    register / hold / release / unregister / query operations with blocks coming and going ~3.6e4 times (tests/csrc/registry_tsan.cpp).
    No sanitizer report, no violated invariant.  The same driver must FAIL when a fix of rounds 4-5 is reverted: (1) a register that
    meets a block whose last owner is leaving does not revive it (round 5's lost owner), (2) the last owner unpins although its bounded
    wait found a transfer still in flight."""
    csrc = os.path.join(ROOT, "eppm_amd", "csrc")
    drv = os.path.join(ROOT, "tests", "csrc", "registry_tsan.cpp")

    def build(inc, exe):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", inc, drv, "-o", exe, "-pthread"])
    exe = str(tmp_path / "registry_tsan")
    build(csrc, exe)
    out = subprocess.run([exe, "8", "100000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr and ", 0 errors" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    pins = int(out.stdout.split(" pins")[0].split()[-1])
    assert pins > 1000, out.stdout            # blocks really were given up and registered again
    hdr = open(os.path.join(csrc, "host_registry.h")).read()
    mutations = (("it->second.closing = false;           // the last owner was on its way out", "/* reverted */                        // the last owner was on its way out"),
                 ("if (!idle || it->second.users != 0) {", "if (false) {"))
    for k, (a, b) in enumerate(mutations):
        assert hdr.count(a) == 1, a
        d = tmp_path / f"mut{k}"
        d.mkdir()
        (d / "host_registry.h").write_text(hdr.replace(a, b))
        build(str(d), str(d / "t"))
        bad = subprocess.run([str(d / "t"), "8", "100000"], capture_output=True, text=True, timeout=600)
        assert bad.returncode != 0 and ", 0 errors" not in bad.stdout, (k, bad.stdout[-300:])


def test_drop_in_class_header_compiles_and_links(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "bao_flow_patchmatch_multiscale_cuda.h"\n'
                   "int main(){ bao_flow_patchmatch_multiscale_cuda e; unsigned char*** a=0; float** u=0;\n"
                   " if (0) { e.init(4,4); e.init(a,a,4,4); bool ok=e.set_data(a,a); (void)ok; e.compute_flow(u,u); e.compute_flow(u,u,a);} return 0; }\n")
    libdir = os.path.dirname(eppm_amd.lib_path())
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "t"),
                           "-L", libdir, "-leppm_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    subprocess.check_call([str(tmp_path / "t")])


def test_argument_errors_without_gpu():
    L = eppm_amd.lib()
    ctx = C.c_void_p()
    small = C.create_string_buffer(8)
    assert L.eppm_device_pci_bus_id(0, small, C.c_size_t(8)) == 1 and L.eppm_device_pci_bus_id(0, None, C.c_size_t(32)) == 1      # EPPM_ERR_ARG
    node, n = C.c_int(7), C.c_int(7)
    rc = L.eppm_bind_thread_to_device(0, C.byref(node), C.byref(n))          # no device here: an error code, outputs reset, nothing bound
    assert rc in (0, 2) and (rc == 0 or (node.value, n.value) == (-1, 0))
    assert L.eppm_create(C.byref(ctx), 2, 2, 0, None) == 1            # EPPM_ERR_ARG: too small
    assert b"out of range" in L.eppm_last_error()
    p = eppm_amd.Params(patch_r=40)
    assert L.eppm_create(C.byref(ctx), 100, 100, 0, C.byref(p)) == 1
    assert L.eppm_compute(None, None, None) == 1
    e = eppm_amd.EPPM()
    with pytest.raises(eppm_amd.EppmError):
        e.compute_flow()
    with pytest.raises(TypeError):
        eppm_amd.Params(nonsense=1)


def test_default_params_are_defs_h():
    p = eppm_amd.Params()
    assert (p.patch_r, p.num_iter, p.search_range, p.num_guess, p.seg_len, p.wmf_iters, p.seed, p.propagation, p.levels) == (9, 10, 30, 6, 10, 20, 1234, 0, 3)


def test_flo_roundtrip_and_header(tmp_path):
    rng = np.random.default_rng(0)
    u = rng.standard_normal((7, 13)).astype(np.float32); v = rng.standard_normal((7, 13)).astype(np.float32)
    path = str(tmp_path / "a.flo")
    eppm_amd.io.save_flo(path, u, v)
    raw = open(path, "rb").read()
    assert raw[:4] == b"PIEH" and np.frombuffer(raw[4:12], np.int32).tolist() == [13, 7] and len(raw) == 12 + 7 * 13 * 8
    u2, v2 = eppm_amd.io.load_flo(path)
    assert (u2 == u).all() and (v2 == v).all()
    with pytest.raises(eppm_amd.EppmError):
        eppm_amd.io.save_flo(str(tmp_path / "a.txt"), u, v)       # extension .flo required (flowIO.cpp:131)


needs_ref = pytest.mark.skipif(O.refio() is None, reason="oracle/_ref not built (reference sources absent)")


@needs_ref
def test_flo_writer_equals_reference_bytes(tmp_path):
    R = O.refio()
    rng = np.random.default_rng(1)
    u = (rng.standard_normal((20, 31)) * 30).astype(np.float32); v = (rng.standard_normal((20, 31)) * 30).astype(np.float32)
    u[3, 4] = v[3, 4] = 1e10
    ours, ref = str(tmp_path / "o.flo"), str(tmp_path / "r.flo")
    eppm_amd.io.save_flo(ours, u, v)
    R.refio_save_flo(ref.encode(), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), 20, 31)
    assert open(ours, "rb").read() == open(ref, "rb").read()
    ru, rv = np.zeros_like(u), np.zeros_like(v)
    assert R.refio_load_flo(ours.encode(), ru.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p), 20, 31) == 0
    assert (ru == u).all() and (rv == v).all()


@needs_ref
def test_ppm_reader_equals_reference(frames):
    R = O.refio()
    path = os.path.join(GOLDEN, "frame10.ppm")
    ours = eppm_amd.io.load_ppm(path)
    ref = np.zeros((480, 640, 3), np.uint8)
    assert R.refio_load_ppm(path.encode(), ref.ctypes.data_as(C.c_void_p), 480, 640) == 3
    assert (ours == ref).all() and (ours == frames[0]).all()
    assert eppm_amd.io.ppm_size(path) == (480, 640)


@needs_ref
def test_epe_aae_equal_reference():
    R = O.refio()
    rng = np.random.default_rng(2)
    shp = (24, 40)
    u, v, gu, gv = [(rng.standard_normal(shp) * 5).astype(np.float32) for _ in range(4)]
    gu[0, :5] = 0; gv[0, :5] = 0          # zero ground truth is skipped by the reference's rule
    gu[1, 1] = 1e10
    epe, aae = C.c_float(), C.c_float()
    R.refio_flow_error(*[a.ctypes.data_as(C.c_void_p) for a in (u, v, gu, gv)], 24, 40, C.byref(epe), C.byref(aae))
    e2, a2 = eppm_amd.io.flow_error(u, v, gu, gv)
    assert e2 == epe.value and a2 == aae.value


def _flow_fields(seed, shp=(37, 53)):
    rng = np.random.default_rng(seed)
    u, v, gu, gv = [(rng.standard_normal(shp) * 6).astype(np.float32) for _ in range(4)]
    gu[0, :5] = 0; gv[0, :5] = 0                 # zero ground truth
    gu[3, 7] = 1e10; gv[5, 9] = -1e10            # unknown ground truth in one component
    gu[8, 8] = 1e10; gv[8, 8] = 1e10             # ... in both
    u[10, 10] = 1e10; v[11, 11] = 2e9            # unknown estimated vectors
    return u, v, gu, gv


@needs_ref
@pytest.mark.parametrize("border", [0, 1, 5, 18, 30])
def test_flow_error_with_border_equals_reference(border):
    """eppm_flow_error_border vs the reference's compiled bao_calc_flow_error (basic/bao_flow_tools.cpp:64-111), bit for bit, including
    a border wider than half the image (no pixel counted: the outputs stay as passed in the reference, 0 here)."""
    R = O.refio()
    u, v, gu, gv = _flow_fields(11)
    h, w = u.shape
    epe, aae = C.c_float(0), C.c_float(0)
    R.refio_flow_error_border(*[a.ctypes.data_as(C.c_void_p) for a in (u, v, gu, gv)], h, w, border, C.byref(epe), C.byref(aae))
    assert eppm_amd.io.flow_error(u, v, gu, gv, border=border) == (epe.value, aae.value)


@needs_ref
@pytest.mark.parametrize("thresh", [0, 1, 3, 10])
def test_flow_error_percentage_equals_reference(thresh):
    """eppm_flow_error_percentage vs bao_calc_flow_error_percentage (:114-141): the fraction bit for bit, the error map byte for byte."""
    R = O.refio()
    R.refio_flow_error_percentage.restype = C.c_float
    u, v, gu, gv = _flow_fields(12)
    h, w = u.shape
    m = np.full((h, w), 7, np.uint8)
    want = R.refio_flow_error_percentage(*[a.ctypes.data_as(C.c_void_p) for a in (u, v, gu, gv)], h, w, thresh, m.ctypes.data_as(C.c_void_p))
    got, gm = eppm_amd.io.flow_error_percentage(u, v, gu, gv, thresh, want_map=True)
    assert got == want and np.array_equal(gm, m)
    assert eppm_amd.io.flow_error_percentage(u, v, gu, gv, thresh) == want
    allunk = np.full((h, w), 1e10, np.float32)
    assert eppm_amd.io.flow_error_percentage(u, v, allunk, allunk, thresh) == 0.0


@needs_ref
@pytest.mark.parametrize("cutoff,cut_invalid", [(4, False), (4, True), (-7, False), (0, True), (1000, False)])
def test_flow_cutoff_equals_reference(cutoff, cut_invalid):
    """eppm_flow_cutoff vs bao_flow_cutoff (:166-197), bit for bit (negative cut-off values, unknown vectors kept or cut)."""
    R = O.refio()
    u, v, _, _ = _flow_fields(13)
    h, w = u.shape
    wu, wv = np.empty_like(u), np.empty_like(v)
    R.refio_flow_cutoff(wu.ctypes.data_as(C.c_void_p), wv.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                        h, w, cutoff, int(cut_invalid))
    gu_, gv_ = eppm_amd.io.flow_cutoff(u, v, cutoff, cut_invalid)
    assert np.array_equal(gu_.view(np.uint32), wu.view(np.uint32)) and np.array_equal(gv_.view(np.uint32), wv.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("seed,scale", [(1, 6.0), (2, 0.01), (3, 300.0)])
def test_host_flow_colour_coding_equals_reference(seed, scale):
    """eppm_flow_to_color_host vs the reference's compiled bao_convert_flow_to_colorshow (:200-231) on Middlebury's computeColor
    (3rdparty/middlebury/colorcode.cpp): byte for byte, unknown vectors black, scaled by the largest known radius."""
    R = O.refio()
    rng = np.random.default_rng(seed)
    h, w = 61, 83
    u = (rng.standard_normal((h, w)) * scale).astype(np.float32)
    v = (rng.standard_normal((h, w)) * scale).astype(np.float32)
    u[0, :] = 0; v[0, :] = np.linspace(-scale, scale, w, dtype=np.float32)       # axis-aligned vectors
    v[1, :] = 0; u[1, :] = np.linspace(-scale, scale, w, dtype=np.float32)
    u[2, 2] = v[2, 2] = 0
    u[5, 5] = 1e10; v[6, 6] = -3e9
    want = np.empty((h, w, 3), np.uint8)
    R.refio_flow_to_color(want.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), h, w)
    got = eppm_amd.io.flow_to_color(u, v)
    bad = np.argwhere((got != want).any(axis=2))
    assert len(bad) == 0, (len(bad), bad[:5], got[tuple(bad[0])] if len(bad) else None, want[tuple(bad[0])] if len(bad) else None)
    assert (got[5, 5] == 0).all() and (got[6, 6] == 0).all()


def test_host_io_under_sanitizers(tmp_path):
    """eppm_io.cpp (PPM / .flo readers, flow tools: no GPU code) under ASan + UBSan + float-cast checks: malformed and truncated
    files, absurd header sizes, static / all-unknown / NaN flow fields, INT_MIN cutoff (tests/csrc/io_sanitize_driver.cpp)."""
    exe = str(tmp_path / "io_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "csrc", "io_sanitize_driver.cpp"), os.path.join(ROOT, "eppm_amd", "csrc", "eppm_io.cpp"), "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_colour_coding_of_a_static_scene_is_white_not_a_crash():
    """maxrad == 0 divides 0 by 0 in the reference (bao_flow_tools.cpp:229, then colorwheel[(int)NaN]): undefined there; here the
    scale falls back to 1 as in Middlebury's color_flow tool, and NaN vectors are drawn black like unknown ones."""
    from eppm_amd import io
    z = np.zeros((6, 9), np.float32)
    assert (io.flow_to_color(z, z) == 255).all()
    u = z.copy(); u[2, 3] = np.nan; u[4, 4] = 3.0
    rgb = io.flow_to_color(u, z)
    assert (rgb[2, 3] == 0).all() and (rgb[0, 0] == 255).all() and not (rgb[4, 4] == 255).all()


def test_reference_main_cpp_builds_unmodified_on_the_drop_in():
    """oracle/_ref/runeppm_ref = the reference's own main.cpp (+ its host-only I/O sources), compiled unmodified against
    include/ and linked with libeppm_hip.so (oracle/Makefile, target runeppm_ref).  Here: it was built and it resolves the
    library; the GPU suite runs it and compares its flow.flo with runeppm's."""
    if not os.path.isdir("/root/reference") and O.runeppm_ref() is None:
        pytest.skip("reference sources absent and no prebuilt oracle/_ref/runeppm_ref")
    O.build()
    exe = O.runeppm_ref()
    assert exe is not None
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libeppm_hip.so" in out and "not found" not in out.split("libeppm_hip.so")[1].splitlines()[0], out
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for s in ("bao_flow_patchmatch_multiscale_cuda", "bao_timer_gpu_cpu"):
        assert s in syms, s                            # the class and the timer come from the drop-in library


def test_timers_header_is_plain_cxx(tmp_path):
    """include/bao_basic_cuda.h needs no GPU runtime header (the reference's pulls in cuda_runtime.h)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "bao_basic_cuda.h"\nint main(){ if (0) { bao_timer_gpu_cpu t; t.start(); t.time_display("x"); bao_timer_gpu g; g.start(); g.stop(); } return 0; }\n')
    libdir = os.path.dirname(eppm_amd.lib_path())
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "t"),
                           "-L", libdir, "-leppm_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])


def test_device_code_uses_global_not_flat_memory_instructions(tmp_path):
    """Every device pointer of the kernels derives from a kernel argument, so the compiler must be able to prove the global
    address space: a flat_load / flat_store / flat_atomic in the gfx950 code object means some pointer arithmetic lost it
    (as a round trip through an integer did once: 8 % slower end to end) -- and it also defeats the wave-aggregated atomics."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not installed")
    so = tmp_path / "lib.so"
    shutil.copy(eppm_amd.lib_path(), so)
    subprocess.run([objdump, "--offloading", str(so)], capture_output=True, text=True, check=True)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert cos, os.listdir(tmp_path)
    n_global = n_flat = 0
    for co in cos:
        dis = subprocess.run([objdump, "-d", str(tmp_path / co)], capture_output=True, text=True, check=True).stdout
        n_global += len(re.findall(r"\bglobal_(?:load|store|atomic)", dis))
        n_flat += len(re.findall(r"\bflat_(?:load|store|atomic)", dis))
    assert n_global > 500 and n_flat == 0, (n_global, n_flat)


def test_delta_index_covers_every_byte_pair():
    """The two-level index behind the range terms that are read from a table (eppm_device.cuh: DeltaTab; api_common.cpp: delta_index),
    restated in numpy: the L-inf distance of two unorm8 texels is |fl(a/255) - fl(b/255)| for two bytes -- 598 distinct floats --, the
    first level kd = round(d * 255) (as trunc(fma(d, 1020, 2)) >> 2 and as the bits of the denormal product d * (1020 * 2^-149), the kernels'
    form) equals |a - b| for every pair, the floats of one kd span at most 65 consecutive bit
    patterns, and the spans add up to what the kernels reserve (kDeltaSlots)."""
    g = (np.arange(256, dtype=np.float32) / np.float32(255)).astype(np.float32)
    a, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    d = np.abs(g[a] - g[b]).astype(np.float32)
    assert len(np.unique(d)) == 598
    kd = np.abs(a - b)
    first = (d.astype(np.float64) * 1020.0 + 2.0).astype(np.float32).astype(np.int64) >> 2        # one rounding: the fma
    assert np.array_equal(first, kd)
    # the form the kernels use: ONE float operation whose result is denormal -- its bits are round(d * 1020) = 4 * kd (IEEE, denormals kept;
    # the kernels add the table's LDS address in the same fma: an integer, the rounding is the product's)
    denorm = (d * np.array([1020], dtype=np.uint32).view(np.float32)[0]).astype(np.float32).view(np.uint32).astype(np.int64)
    assert np.array_equal(denorm, 4 * kd)
    bits = d.view(np.uint32).astype(np.int64)
    spans = [int(bits[kd == k].max() - bits[kd == k].min() + 1) for k in range(256)]
    slots = int(re.search(r"constexpr int kDeltaSlots = (\d+);", open(os.path.join(ROOT, "eppm_amd", "csrc", "eppm_internal.h")).read()).group(1))
    assert max(spans) == 65 and sum(spans) == 799 <= slots
    assert float(d.max()) == 1.0            # the smoothing clamps its sentinel taps to this distance, whose weight is exp(-2500) = 0
