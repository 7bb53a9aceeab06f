import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The oracle's lockstep sweeps synchronise every step: on the GPU box (256 hardware threads) OpenMP's default of one
# thread per CPU makes a 160x120 run take 19 s instead of 0.15 s with 16 threads (tools/orc_threads.py).  Set before
# libgomp loads; child processes (bench.py, the CLI tests) inherit it.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))


# The pytest process loads libeppm_hip_test.so: the product library's own objects plus the switches and probes of include/eppm_test.h
# (eppm_amd/csrc/Makefile).  In-process only: bench.py, runeppm, the reference's main.cpp and smoke(), which the tests start as child
# processes, load the product library.
import eppm_amd  # noqa: E402

eppm_amd.select_library("test")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_ppm(path):
    """Minimal P6 reader for the test fixtures (comment lines tolerated)."""
    b = open(path, "rb").read()
    toks, i = [], 0
    while len(toks) < 4:
        j = b.index(b"\n", i)
        line, i = b[i:j], j + 1
        if not line.startswith(b"#"):
            toks += line.split()
    w, h = int(toks[1]), int(toks[2])
    return np.frombuffer(b[i:i + w * h * 3], np.uint8).reshape(h, w, 3).copy()


@pytest.fixture(scope="session")
def frames():
    """The reference's bundled Middlebury pair (frame10/frame11.ppm, 640x480)."""
    return read_ppm(os.path.join(GOLDEN, "frame10.ppm")), read_ppm(os.path.join(GOLDEN, "frame11.ppm"))


@pytest.fixture(scope="session")
def crop(frames):
    """160x120 centre crop of the bundled pair: the oracle finishes the whole path in ~1 s."""
    a, b = frames
    return a[180:300, 240:400].copy(), b[180:300, 240:400].copy()


@pytest.fixture(scope="session")
def crop_stages(crop):
    """Oracle run of the whole path on the crop with every intermediate plane."""
    from oracle import oracle as O
    u, v, st = O.compute_flow(crop[0], crop[1], dump=True)
    st["u"], st["v"] = u, v
    return st
