"""oracle/cloudgen.c pinned by the reference's own source: PointCloudMapping::InitCamGridPoints and
::GeneratePointCloudInCameraFrameBGRA, the two definitions cut verbatim out of /root/reference/src/PointCloudMapping.cc at
build time and compiled against stand-ins (oracle/ref/cloudgen_ref_wrap.cpp says which) into oracle/_ref/libcloudgen_ref.so.

  CPU  the oracle equals the compiled reference: grid table, every byte of every point record, pixelToPointIndex —
       images with holes (no return, NaN, beyond range), steps 1 / 2 / 3 / 4, odd sizes, padded rows, limits that cut;
       the committed reference-made digests (tests/golden/cloudgen_reference_digests.json) are reproduced by the oracle
       alone (this half runs where oracle/_ref is absent too).
  GPU  the HIP path reproduces the reference-made digests through the C ABI."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import cloudgen_golden_scenario as S
from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libcloudgen_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "cloudgen_reference_digests.json")
_i, _d, _vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


_ref_lib = [None]


def _ref():
    if _ref_lib[0] is None:
        lib = ctypes.CDLL(REF)
        lib.ref_cloudgen_create.restype = _vp
        lib.ref_cloudgen_create.argtypes = [_vp, _vp, _i, _d, _i, _d, _d]
        lib.ref_cloudgen_destroy.argtypes = [_vp]
        lib.ref_cloudgen_grid.argtypes = [_vp, _i, _i, _vp, _i]
        lib.ref_cloudgen_generate.argtypes = [_vp, _vp, _i, _vp, _i, _i, _i, ctypes.c_ulong, _d, _vp, _i, _vp, _vp]
        assert lib.ref_cloudgen_point_size() == 48
        _ref_lib[0] = lib
        # GeneratePointCloudInCameraFrameBGRA sizes its index table by a function-local STATIC (src/PointCloudMapping.cc:944:
        # N of the first image it ever sees): the first call of the process goes to the largest grid of these tests
        h = _vp(lib.ref_cloudgen_create(S.K_TUM.ctypes.data, np.zeros(5, np.float32).ctypes.data, 5, 40.0, 1, 0.1, 5.0))
        d, c = np.zeros((480, 640), np.float32), np.zeros((480, 640, 3), np.uint8)
        out = np.zeros(640 * 480 * 48, np.uint8)
        lib.ref_cloudgen_generate(h, d.ctypes.data, 640, c.ctypes.data, 1920, 640, 480, 0, 0.0, out.ctypes.data, 640 * 480, None, None)
        lib.ref_cloudgen_destroy(h)
    return _ref_lib[0]


def ref_generator(width, height, step, K, min_depth, max_depth, depth_pitch=None, bgr_pitch=None):
    lib = _ref()
    K = np.ascontiguousarray(K, np.float32)
    h = _vp(lib.ref_cloudgen_create(K.ctypes.data, np.zeros(5, np.float32).ctypes.data, 5, 40.0, step, min_depth, max_depth))
    ngrid = ((width + step - 1) // step) * ((height + step - 1) // step)
    grid = np.zeros((ngrid, 2), np.float32)
    assert lib.ref_cloudgen_grid(h, width, height, grid.ctypes.data, ngrid) == ngrid

    def gen(depth, bgr, kfid):
        out = np.zeros(ngrid, oracle_lib.SURFEL_DTYPE)
        p2p = np.zeros((height, width), np.int32)
        n = lib.ref_cloudgen_generate(h, depth.ctypes.data, depth_pitch or depth.shape[1], bgr.ctypes.data,
                                      bgr_pitch or 3 * bgr.shape[1], width, height, kfid, 12.5, out.ctypes.data, ngrid,
                                      p2p.ctypes.data, None)
        assert 0 <= n <= ngrid
        return out[:n], p2p
    gen.handle = h      # (kept alive by the closure; a handful per process)
    return grid, gen


def oracle_generator(oracle):
    def make(width, height, step, K, min_depth, max_depth, depth_pitch=None, bgr_pitch=None):
        K = np.asarray(K, np.float32)
        # the reference's fx, fy, cx, cy are doubles holding the FLOAT entries of K (src/PointCloudMapping.cc:177-180)
        grid = oracle.cam_grid_points(width, height, step, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))

        def gen(depth, bgr, kfid):
            return oracle.cloudgen(depth, bgr, grid, step, min_depth, max_depth, kfid, depth_pitch=depth_pitch,
                                   bgr_pitch=bgr_pitch, width=width, height=height)
        return grid, gen
    return make


needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libcloudgen_ref.so is built where /root/reference exists")


@needs_ref
@pytest.mark.parametrize("case", S.CASES, ids=[c["id"] for c in S.CASES])
def test_oracle_equals_the_compiled_reference(oracle, case):
    args = (case["width"], case["height"], case["step"], S.K_TUM, case["min_depth"], case["max_depth"])
    rgrid, rgen = ref_generator(*args)
    ogrid, ogen = oracle_generator(oracle)(*args)
    assert rgrid.tobytes() == ogrid.tobytes()
    for k, (depth, bgr) in enumerate(S.frames(case, 3)):
        want, want_p2p = rgen(depth, bgr, 7 + k)
        got, got_p2p = ogen(depth, bgr, 7 + k)
        assert len(got) == len(want) > 500
        assert np.asarray(got).tobytes() == np.asarray(want).tobytes()       # x, y, z, kfid, normal, colour, depth, labels: every byte
        assert np.array_equal(got_p2p, want_p2p)


@needs_ref
def test_oracle_equals_the_compiled_reference_on_padded_rows_and_degenerate_images(oracle):
    w, h, step = 320, 200, 2
    rng = np.random.default_rng(5)
    rgrid, rgen = ref_generator(w, h, step, S.K_TUM, 0.1, 5.0, depth_pitch=w + 24, bgr_pitch=3 * w + 40)
    ogrid, ogen = oracle_generator(oracle)(w, h, step, S.K_TUM, 0.1, 5.0, depth_pitch=w + 24, bgr_pitch=3 * w + 40)
    for fill in ("noise", 0.0, np.nan, 5.0, np.float32(0.1), 2.0, "one"):
        depth = np.full((h, w + 24), 777.0, np.float32)
        if fill == "noise":
            depth[:, :w] = rng.random((h, w), dtype=np.float32) * 7.0
        elif fill == "one":
            depth[:, :w] = 0.0
            depth[h - 2, w - 2] = 1.0
        else:
            depth[:, :w] = fill
        bgr = rng.integers(0, 256, (h, 3 * w + 40), dtype=np.uint8)
        want, want_p2p = rgen(depth, bgr, 3)
        got, got_p2p = ogen(depth, bgr, 3)
        assert len(got) == len(want)
        assert np.asarray(got).tobytes() == np.asarray(want).tobytes()
        assert np.array_equal(got_p2p, want_p2p)


def test_oracle_reproduces_the_reference_made_digests(oracle):
    with open(GOLDEN) as f:
        want = json.load(f)["cases"]
    assert S.run(oracle_generator(oracle)) == want


@pytest.mark.gpu
def test_hip_reproduces_the_reference_made_digests():
    from plvs_amd import cloudgen

    def make(width, height, step, K, min_depth, max_depth):
        K = np.asarray(K, np.float32)
        grid = cloudgen.InitCamGridPoints(width, height, step, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
        g = cloudgen.PointCloudGenerator(width, height, grid, step=step, min_depth=min_depth, max_depth=max_depth)
        return grid, lambda depth, bgr, kfid: g.GeneratePointCloudInCameraFrameBGRA(bgr, depth, kfid)
    with open(GOLDEN) as f:
        want = json.load(f)["cases"]
    assert S.run(make) == want
