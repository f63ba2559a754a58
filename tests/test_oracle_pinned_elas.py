"""libelas, the dense stereo matcher behind PointCloudKeyFrame::ProcessStereoLibelas (src/PointCloudKeyFrame.cc:335), is
the one component of the path's neighbourhood for which the reference tree STORES outputs:
Thirdparty/libelas-gpu/GPU_test/2016_12_06_cpu/*_disp.pgm for Thirdparty/libelas-gpu/input/*.pgm.  The CPU sources compile
unmodified (oracle/ref/Makefile -> oracle/_ref/libelas_ref.so); this test runs them over the reference's own inputs and
compares with the stored files as main_cpu.cpp wrote them (disparities scaled to 255 / max).  The stored files come from
another build of the library: validity masks are reproduced exactly, disparities to within one grey level for >= 98.5 %
of the pixels.  Dev-container test: it reads the inputs where they lie under /root/reference (they are not committed:
four 1-MB images per pair) and is skipped where the tree or the library is absent.  The stages of libelas that run on
the device are checked in tests/test_elas.py; nothing on the GPU is checked here.  (The library is built with
oracle/ref/elas_zero_malloc.h in front of the sources: memory they read without having written it is zero, as in a fresh
process.)"""
import ctypes
import os

import numpy as np
import pytest

from tests.pgm import read_pgm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libelas_ref.so")
TREE = "/root/reference/Thirdparty/libelas-gpu"

pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir(TREE)),
                                reason="needs oracle/_ref/libelas_ref.so and the reference tree's stored libelas outputs")


@pytest.mark.parametrize("name", ["cones", "raindeer"])
def test_compiled_libelas_against_the_outputs_the_reference_stores(name):
    lib = ctypes.CDLL(REF)
    lib.ref_elas_process.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
    left, right = read_pgm(f"{TREE}/input/{name}_left.pgm"), read_pgm(f"{TREE}/input/{name}_right.pgm")
    h, w = left.shape
    d1, d2 = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    lib.ref_elas_process(left.ctypes.data, right.ctypes.data, w, h, w, 0, 0, d1.ctypes.data, d2.ctypes.data)
    dmax = float(max(d1.max(), d2.max()))
    for d, side in ((d1, "left"), (d2, "right")):
        out = np.maximum(255.0 * d.astype(np.float64) / dmax, 0.0).astype(np.uint8)      # main_cpu.cpp:76-79
        stored = read_pgm(f"{TREE}/GPU_test/2016_12_06_cpu/{name}_{side}_disp.pgm")
        assert np.array_equal(out > 0, stored > 0), f"{name} {side}: validity masks differ"
        close = np.abs(out.astype(int) - stored.astype(int)) <= 1
        assert close.mean() >= 0.985, f"{name} {side}: {1 - close.mean():.3%} of the pixels off by more than one level"
