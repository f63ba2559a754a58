"""The LSD line detector (Line.LSD.on: 1 -> LineExtractor::skUseLsdExtractor):
cv::lsd::LineSegmentDetectorImpl::detect (Thirdparty/line_descriptor/src/lsd_custom.cpp), LSDDetectorC::detect
(LSDDetector_custom.cpp:50-298) and LineExtractor::operator() with that detector (src/LineExtractor.cc:199-289).

The checker of this row is the REFERENCE ITSELF: those sources compiled unmodified against the OpenCV stand-in
(oracle/_ref/liblsd_ref.so, oracle/ref/lsd_ref_wrap.cpp) and digests they made (tests/golden/lsd_reference_digests.json,
scripts/make_lsd_golden.py); there is no restatement of LSD under oracle/.

  CPU  the product's host stages (tap tables, the unstable ordering, region growing, refinement, NFA: plvs_amd/csrc/lsd_host.hpp,
       compiled by g++ behind plain-loop versions of the three device kernels, tests/host/lsd_host.cpp) == the compiled
       reference, segment for segment, bit for bit; == the committed digests where the reference is absent
  GPU  the HIP path through the C ABI == the compiled reference and the digests: segments, KeyLines, descriptors
Bar: bit-exact (every float of every record)."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import lsd_golden_scenario as S
from tests import oracle_lib
from tests.oracle_lib import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "liblsd_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "lsd_reference_digests.json")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/liblsd_ref.so is built where /root/reference exists")

_vp, _i, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
_OPT_KEYS = ["refine", "scale", "sigma_scale", "quant", "ang_th", "log_eps", "density_th", "n_bins"]
_OPT_TYPES = [_i, _d, _d, _d, _d, _d, _d, _i]


def _opt_args(o):
    return [int(o[k]) if t is _i else float(o[k]) for k, t in zip(_OPT_KEYS, _OPT_TYPES)]


def _keyline_dtype():
    from plvs_amd.lines import KEYLINE_DTYPE
    return KEYLINE_DTYPE


class RefBackend:
    """The reference's compiled sources."""

    def __init__(self):
        self.lib = ctypes.CDLL(REF)
        self.lib.ref_lsd_segments.argtypes = [_vp, _i, _i, _i] + _OPT_TYPES + [_vp, _i]
        self.lib.ref_lsd_detect.argtypes = [_vp, _i, _i, _i, _i, ctypes.c_float] + _OPT_TYPES + [_d, _vp, _i]
        self.lib.ref_lsd_extract.argtypes = [_vp, _i, _i, _i, _i, _i] + _OPT_TYPES + [_d, _d, _vp, _vp, _i]

    def segments(self, image, **o):
        h, w = image.shape
        out = np.zeros((65536, 4), np.float32)
        n = self.lib.ref_lsd_segments(image.ctypes.data, w, h, image.strides[0], *_opt_args(o), out.ctypes.data, len(out))
        assert 0 <= n <= len(out)
        return out[:n].copy()

    def detect(self, image, num_octaves, pyramid_scale, o, min_length):
        h, w = image.shape
        kl = np.zeros(65536, _keyline_dtype())
        n = self.lib.ref_lsd_detect(image.ctypes.data, w, h, image.strides[0], num_octaves, float(pyramid_scale), *_opt_args(o),
                                    float(min_length), kl.ctypes.data, len(kl))
        assert 0 <= n <= len(kl)
        return kl[:n].copy()

    def extract(self, image, nfeatures, num_octaves, o, min_length):
        h, w = image.shape
        kl = np.zeros(65536, _keyline_dtype())
        d = np.zeros((65536, 32), np.uint8)
        n = self.lib.ref_lsd_extract(image.ctypes.data, w, h, image.strides[0], nfeatures, num_octaves, *_opt_args(o),
                                     float(min_length), 1.6, kl.ctypes.data, d.ctypes.data, len(kl))
        assert 0 <= n <= len(kl)
        return kl[:n].copy(), d[:n].copy()


class HostBackend:
    """plvs_amd/csrc/lsd_host.hpp on the CPU (segments only)."""

    def __init__(self):
        src = os.path.join(oracle_lib.HOSTCORE_DIR, "lsd_host.cpp")
        hdr = os.path.join(ROOT, "plvs_amd", "csrc", "lsd_host.hpp")
        self.lib = ctypes.CDLL(oracle_lib._host_build("libhostlsd", src, [hdr]))
        self.lib.hostlsd_segments.argtypes = [_vp, _i, _i, _i] + _OPT_TYPES + [_vp, _i]

    def segments(self, image, **o):
        h, w = image.shape
        out = np.zeros((65536, 4), np.float32)
        n = self.lib.hostlsd_segments(image.ctypes.data, w, h, image.strides[0], *_opt_args(o), out.ctypes.data, len(out))
        assert 0 <= n <= len(out)
        return out[:n].copy()


class HipBackend:
    """The product, through the Python mirror of the reference's interfaces (plvs_amd/lines.py) over the C ABI."""

    def segments(self, image, **o):
        from plvs_amd.lines import createLineSegmentDetector
        det = createLineSegmentDetector(o["refine"], o["scale"], o["sigma_scale"], o["quant"], o["ang_th"], o["log_eps"],
                                        o["density_th"], o["n_bins"])
        out = det.detect(image)
        det.close()
        return out

    def detect(self, image, num_octaves, pyramid_scale, o, min_length):
        from plvs_amd.lines import LSDDetectorC, LSDOptions
        opts = LSDOptions(numOctaves=num_octaves, min_length=min_length, **o)
        det = LSDDetectorC.createLSDDetectorC(opts)
        kl = det.detect(image, pyramid_scale, num_octaves, opts)
        det.close()
        return kl

    def extract(self, image, nfeatures, num_octaves, o, min_length):
        from plvs_amd.lines import LineExtractor, LSDOptions

        class Lsd(LineExtractor):
            skUseLsdExtractor = True
        ex = Lsd(nfeatures, LSDOptions(numOctaves=num_octaves, min_length=min_length, **o))
        kl, d = ex(image)
        ex.close()
        return kl, d


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


# ------------------------------------------------------------------------------------------------ CPU

@needs_ref
@pytest.mark.parametrize("case", S.SEGMENT_CASES, ids=[c["id"] for c in S.SEGMENT_CASES])
def test_host_stages_equal_the_compiled_reference(case):
    img, o = S.image_of(case), S.options(case)
    want, got = RefBackend().segments(img, **o), HostBackend().segments(img, **o)
    assert len(want) == len(got) > 100
    assert want.tobytes() == got.tobytes()


@needs_ref
def test_host_stages_equal_the_compiled_reference_on_degenerate_images():
    rng = np.random.default_rng(3)
    ref, host = RefBackend(), HostBackend()
    flat = np.full((120, 160), 90, np.uint8)
    noise = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    step = np.zeros((64, 200), np.uint8)
    step[:, 100:] = 255                                             # one vertical edge
    diag = (np.add.outer(np.arange(150), np.arange(150)) > 150).astype(np.uint8) * 200
    small = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    padded = np.ascontiguousarray(rng.integers(0, 256, (80, 128), dtype=np.uint8))[:, :101]   # stride > width
    for img in (flat, noise, step, diag, small, padded):
        for o in (S.DEFAULTS, dict(S.DEFAULTS, **S.TRACKING), dict(S.DEFAULTS, scale=1.0, refine=0)):
            want, got = ref.segments(img, **o), host.segments(img, **o)
            assert len(want) == len(got) and want.tobytes() == got.tobytes()
    assert len(ref.segments(flat, **S.DEFAULTS)) == 0 and len(ref.segments(step, **S.DEFAULTS)) >= 1


@pytest.mark.parametrize("sort_threads", ["0", "1"], ids=["one_thread_sort", "partitions_on_threads"])
def test_host_stages_reproduce_the_reference_made_digests(sort_threads, monkeypatch):
    """The ordering is std::sort's on one thread, or libstdc++'s own partition steps with the right halves on other threads
    (lsd_host.hpp, sort_as_std): the same permutation either way, i.e. the same segments in the same order."""
    monkeypatch.setenv("PLVS_LSD_SORT_THREADS", sort_threads)
    assert S.run(HostBackend(), parts=("segments",))["segments"] == _golden()["segments"]


@needs_ref
def test_compiled_reference_reproduces_its_digests():
    assert S.run(RefBackend()) == _golden()


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_hip_reproduces_the_reference_made_digests():
    assert S.run(HipBackend()) == _golden()


@pytest.mark.gpu
@needs_ref
def test_hip_equals_the_compiled_reference_record_by_record():
    ref, hip = RefBackend(), HipBackend()
    for c in S.SEGMENT_CASES:
        img, o = S.image_of(c), S.options(c)
        want, got = ref.segments(img, **o), hip.segments(img, **o)
        assert len(want) == len(got) > 100, c["id"]
        assert want.tobytes() == got.tobytes(), c["id"]
    for c in S.DETECT_CASES:
        img, o = S.image_of(c), S.options(c)
        want, got = ref.detect(img, c["num_octaves"], c["pyramid_scale"], o, c["min_length"]), \
            hip.detect(img, c["num_octaves"], c["pyramid_scale"], o, c["min_length"])
        assert len(want) == len(got) > 20, c["id"]
        for f in want.dtype.names:
            assert np.array_equal(want[f].view(np.uint32), got[f].view(np.uint32)), (c["id"], f)
    for c in S.EXTRACT_CASES:
        img, o = S.image_of(c), S.options(c)
        wk, wd = ref.extract(img, c["nfeatures"], c["num_octaves"], o, c["min_length"])
        gk, gd = hip.extract(img, c["nfeatures"], c["num_octaves"], o, c["min_length"])
        assert len(wk) == len(gk) > 20, c["id"]
        for f in wk.dtype.names:
            assert np.array_equal(wk[f].view(np.uint32), gk[f].view(np.uint32)), (c["id"], f)
        assert np.array_equal(wd, gd), c["id"]


@pytest.mark.gpu
def test_hip_lsd_edge_cases():
    from plvs_amd import _lib
    from plvs_amd.lines import LSDDetectorC, LSDOptions, LineExtractor, createLineSegmentDetector
    det = createLineSegmentDetector()
    assert len(det.detect(np.full((120, 160), 90, np.uint8))) == 0           # a flat image: no level line is defined
    step = np.zeros((64, 200), np.uint8)
    step[:, 100:] = 255
    s = det.detect(step)
    assert len(s) >= 1 and np.all(np.abs(s[:, 0] - s[:, 2]) < 1.0)          # vertical segments along the edge
    # one handle, images of different sizes one after another; a non-contiguous view goes through a copy
    rng = np.random.default_rng(0)
    a = det.detect(S.image_of(S.SEGMENT_CASES[1]))
    det.detect(rng.integers(0, 256, (50, 70), dtype=np.uint8))
    b = det.detect(S.image_of(S.SEGMENT_CASES[1]))
    assert a.tobytes() == b.tobytes() and len(a) > 100
    det.close()
    # the image in device memory, rows further apart than they need to be: the same lines
    import torch

    class LsdDev(LineExtractor):
        skUseLsdExtractor = True
    c0 = S.EXTRACT_CASES[0]
    exd = LsdDev(c0["nfeatures"], LSDOptions(numOctaves=c0["num_octaves"], min_length=c0["min_length"], **S.options(c0)))
    img = S.image_of(c0)
    wide = torch.zeros((img.shape[0], img.shape[1] + 64), dtype=torch.uint8, device="cuda")
    wide[:, :img.shape[1]] = torch.from_numpy(img).cuda()
    kh, dh = exd(img)
    kd, dd = exd(wide[:, :img.shape[1]])
    assert len(kh) == 100 and kh.tobytes() == kd.tobytes() and dh.tobytes() == dd.tobytes()
    exd.close()
    # the extractor with LSD on a flat image: no lines, empty outputs
    class Lsd(LineExtractor):
        skUseLsdExtractor = True
    ex = Lsd(100, LSDOptions(numOctaves=3, **dict(S.DEFAULTS, **S.TRACKING)))
    kl, d = ex(np.full((240, 320), 17, np.uint8))
    assert len(kl) == 0 and d.shape == (0, 32)
    with pytest.raises(_lib.PlvsHipError):
        ex(np.zeros((1, 1), np.uint8))
    ex.close()
    d2 = LSDDetectorC.createLSDDetectorC()
    with pytest.raises(_lib.PlvsHipError):
        d2.detect(S.image_of(S.SEGMENT_CASES[0]), 1.2, 9, LSDOptions(numOctaves=9))     # more octaves than the library holds
    d2.close()


@pytest.mark.gpu
@needs_ref
def test_hip_equals_the_compiled_reference_on_degenerate_images():
    """The images of the CPU test of the host stages, through the device: flat, noise, one edge, a diagonal, 9 x 11, and an
    odd size under three option sets."""
    rng = np.random.default_rng(3)
    ref, hip = RefBackend(), HipBackend()
    flat = np.full((120, 160), 90, np.uint8)
    noise = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    step = np.zeros((64, 200), np.uint8)
    step[:, 100:] = 255
    diag = (np.add.outer(np.arange(150), np.arange(150)) > 150).astype(np.uint8) * 200
    small = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    odd = np.ascontiguousarray(rng.integers(0, 256, (333, 517), dtype=np.uint8))
    for img in (flat, noise, step, diag, small, odd):
        for o in (S.DEFAULTS, dict(S.DEFAULTS, **S.TRACKING), dict(S.DEFAULTS, scale=1.0, refine=0)):
            want, got = ref.segments(img, **o), hip.segments(img, **o)
            assert len(want) == len(got) and want.tobytes() == got.tobytes(), (img.shape, o)
    # more than 2^22 pixels: 64-bit ordering keys on the host, a 4.3-megapixel field on the device
    tiles = [golden(n) for n in ("aloe_640x480.pgm", "cones_640x480.pgm", "aloe_640x480_shift.pgm")]
    row = np.concatenate([tiles[0], tiles[1], tiles[2], tiles[1][:, :480]], 1)
    big = np.ascontiguousarray(np.concatenate([row, row[::-1], row[:, ::-1], row[::-1, ::-1][:360]], 0))
    o = dict(S.DEFAULTS, scale=1.0, refine=1)
    want, got = ref.segments(big, **o), hip.segments(big, **o)
    assert len(want) == len(got) > 10000 and want.tobytes() == got.tobytes()


@needs_ref
def test_host_stages_equal_the_compiled_reference_on_an_image_of_more_than_four_megapixels():
    """Beyond 2^22 pixels the ordering's keys are 64 bits wide (bin above a 32-bit pixel index) instead of 32: the same
    permutation, the same segments.  A 2 400 x 1 800 mosaic of the golden images at scale 1."""
    tiles = [golden(n) for n in ("aloe_640x480.pgm", "cones_640x480.pgm", "aloe_640x480_shift.pgm")]
    row = np.concatenate([tiles[0], tiles[1], tiles[2], tiles[1][:, :480]], 1)          # 480 x 2400
    img = np.ascontiguousarray(np.concatenate([row, row[::-1], row[:, ::-1], row[::-1, ::-1][:360]], 0))   # 1800 x 2400
    assert img.shape[0] * img.shape[1] > (1 << 22)
    o = dict(S.DEFAULTS, scale=1.0, refine=1)
    want, got = RefBackend().segments(img, **o), HostBackend().segments(img, **o)
    assert len(want) == len(got) > 10000
    assert want.tobytes() == got.tobytes()
