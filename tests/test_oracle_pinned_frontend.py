"""The front-end oracles pinned by the reference's OWN sources.

oracle/_ref/libfrontend_ref.so = src/ORBextractor.cc (CPU branch), src/LineExtractor.cc and
Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp compiled UNMODIFIED from /root/reference against the
OpenCV stand-in oracle/ref/cv_full (oracle/ref/Makefile).  Everything that is PLVS's own on this path — the per-cell
FAST loop, DistributeOctTree / DivideNode, IC_Angle, computeOrbDescriptor, the lapping-area packing; OctaveKeyLines with
the octave grouping, EdgeDrawing's anchors and smart routing, the least-squares fits, the Helmholtz validation,
detectLineFeatures' sort / border filter / length cut, computeLBD — comes from the reference's source; the OpenCV image
primitives underneath (FAST, resize, GaussianBlur, Sobel, copyMakeBorder, fastAtan2, cvRound, the small float
products: OpenCV is not in the reference tree) are the restatements of oracle/cv_primitives.hpp on both sides.  So the
statement these tests make is: oracle/orb.cpp and oracle/lines.cpp equal the reference "up to the OpenCV primitives".

 * CPU, needs oracle/_ref (built where /root/reference exists; it travels with the snapshot): the restatements equal
   the compiled reference field by field / byte by byte.
 * CPU, always: the restatements reproduce tests/golden/frontend_reference_digests.json, the digests the compiled
   reference produced (scripts/make_frontend_golden.py) — the fallback where oracle/_ref is absent.
 * GPU: the HIP path reproduces the same file through the C ABI.
Nothing here reads /root/reference at run time."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import frontend_golden_scenario as S
from tests import oracle_lib
from tests.oracle_lib import KP_DTYPE, OracleLines, OracleOrb, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libfrontend_ref.so")
DIGESTS = os.path.join(ROOT, "tests", "golden", "frontend_reference_digests.json")
needs_ref = pytest.mark.skipif(not os.path.exists(REF),
                               reason="oracle/_ref/libfrontend_ref.so (built where /root/reference exists) not present")


class _RefNames:
    """The compiled reference exports the oracle's C entry points under ref_*: OracleOrb / OracleLines drive both."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name.startswith("oracle_"):
            name = "ref_" + name[len("oracle_"):]
        return getattr(self._lib, name)


class _RefOrb(OracleOrb):
    def __init__(self, names, nfeatures, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        lib = names
        self.lib, self.nlevels = lib, nlevels
        lib.oracle_orb_create.restype = ctypes.c_void_p
        lib.oracle_orb_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oracle_orb_destroy.argtypes = [ctypes.c_void_p]
        lib.oracle_orb_extract.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2 + \
            [ctypes.c_int, ctypes.c_void_p]
        lib.oracle_orb_features_per_level.argtypes = [ctypes.c_void_p] * 2
        lib.oracle_orb_umax.argtypes = [ctypes.c_void_p] * 2
        lib.oracle_orb_level_size.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_orb_get_level.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self.h = ctypes.c_void_p(lib.oracle_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th))
        self.cap = nfeatures * 2 + 64


def _ref():
    return _RefNames(ctypes.CDLL(REF))


def ref_extractors():
    """(make_orb, make_lines, make_shared) of tests/frontend_golden_scenario.run over the compiled reference."""
    names = _ref()

    def make_shared(img):
        orb, lines = _RefOrb(names, 1000), OracleLines(names)
        f = names.ref_frame_precompute_pyramid
        f.restype = None
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        img = np.ascontiguousarray(img)
        f(orb.h, lines.h, img.ctypes.data, img.shape[1], img.shape[0], img.shape[1])
        return lines.extract(img)

    return (lambda nf: _RefOrb(names, nf).extract, lambda nf: OracleLines(names, nfeatures=nf).extract, make_shared)


def oracle_extractors(oracle):
    def make_shared(img):
        oorb = oracle.orb(1000, 1.2, 8, 20, 7)
        oorb.extract(img)
        ol = oracle.lines()
        ol.set_pyramid([oorb.level(k) for k in range(8)], 3, 1.2)
        return ol.extract(img)

    return (lambda nf: oracle.orb(nf).extract, lambda nf: oracle.lines(nfeatures=nf).extract, make_shared)


def _same_kps(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)) for f in KP_DTYPE.names)


# ------------------------------------------------------------------ CPU: restatement == compiled reference
@needs_ref
def test_orb_constructor_tables_equal_the_reference(oracle):
    names = _ref()
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (1500, 1.4142135, 5), (5000, 1.2, 8)):
        r, o = _RefOrb(names, nf, sf, nl), oracle.orb(nf, sf, nl)
        assert list(r.features_per_level()) == list(o.features_per_level())
        assert list(r.umax()) == list(o.umax())
        t = [np.zeros(nl, np.float32) for _ in range(4)]
        names.ref_orb_scale_tables.argtypes = [ctypes.c_void_p] * 5
        assert names.ref_orb_scale_tables(r.h, *[x.ctypes.data for x in t]) == nl
        s = np.float32(1)
        for k in range(nl):                                  # ORBextractor.cc:457-470 (scaleFactor is a double member)
            assert t[0][k] == s and t[2][k] == np.float32(s * s)
            assert t[1][k] == np.float32(1) / t[0][k] and t[3][k] == np.float32(1) / t[2][k]
            s = np.float32(np.float64(s) * np.float64(np.float32(sf)))


@needs_ref
@pytest.mark.parametrize("name", S.IMAGES)
def test_orb_restatement_equals_the_compiled_reference(oracle, name):
    names = _ref()
    img = S.image(name)
    for nf in S.ORB_FEATURES:
        for lap in {(0, 0), S.LAPPING.get(name, (0, 0))}:
            r, o = _RefOrb(names, nf), oracle.orb(nf)
            rm, rk, rd = r.extract(img, lap)
            om, ok, od = o.extract(img, lap)
            assert rm == om and _same_kps(rk, ok), f"{name} {nf} {lap}: key points differ"
            assert np.array_equal(rd, od), f"{name} {nf} {lap}: descriptor bits differ"
            for lv in range(8):                              # ComputePyramid and the blurred levels
                assert np.array_equal(r.level(lv), o.level(lv)), f"{name}: pyramid level {lv}"
                if np.any(rk["octave"] == lv):               # (a level without key points is not blurred)
                    assert np.array_equal(r.level(lv, True), o.level(lv, True)), f"{name}: blurred level {lv}"


@needs_ref
def test_orb_edge_cases_equal_the_reference(oracle):
    names = _ref()
    rng = np.random.default_rng(5)
    cases = {"flat": np.full((480, 640), 93, np.uint8),                       # no corner anywhere: minThFAST pass, empty
             "noise": rng.integers(0, 256, (240, 320), dtype=np.uint8),       # every cell saturated
             "small": S.image("cones_640x480.pgm")[:120, :160].copy(),        # upper levels have no cells
             "tiny": S.image("cones_640x480.pgm")[:40, :50].copy(),
             "odd": S.image("aloe_640x480.pgm")[:391, :517].copy(),
             "few": S.image("synth3")}
    for label, img in cases.items():
        for nf, sf, nl in ((1000, 1.2, 8), (300, 1.3, 4)):
            r, o = _RefOrb(names, nf, sf, nl), oracle.orb(nf, sf, nl)
            rm, rk, rd = r.extract(img)
            om, ok, od = o.extract(img)
            assert rm == om and _same_kps(rk, ok) and np.array_equal(rd, od), f"{label} {nf}"
    r, o = _RefOrb(names, 1000), oracle.orb(1000)
    assert r.extract(np.zeros((0, 0), np.uint8))[0] == -1 and o.extract(np.zeros((0, 0), np.uint8))[0] == -1


@needs_ref
@pytest.mark.parametrize("name", S.IMAGES)
def test_lines_restatement_equals_the_compiled_reference(oracle, name):
    names = _ref()
    img = S.image(name)
    for nf in (100, 0, 30):
        r, o = OracleLines(names, nfeatures=nf), oracle.lines(nfeatures=nf)
        rk, rd = r.extract(img)
        ok, od = o.extract(img)
        assert len(rk) == len(ok) and rk.tobytes() == ok.tobytes(), f"{name} {nf}: KeyLine records differ"
        assert np.array_equal(rd, od), f"{name} {nf}: LBD bits differ"
        for oc in range(3):
            assert r.octave_size(oc) == o.octave_size(oc)
            assert r.num_in_octave(oc) == o.num_in_octave(oc), f"{name}: segments in octave {oc}"
            for which in ("dx", "dy"):
                assert np.array_equal(r.octave_map(oc, which), o.octave_map(oc, which)), f"{name}: {which} octave {oc}"


@needs_ref
def test_lines_parameters_and_edge_cases_equal_the_reference(oracle):
    names = _ref()
    img = S.image("cones_640x480.pgm")
    for kw in (dict(nfeatures=0, nlevels=2, scale=1.4142135, min_length=0.05),
               dict(nfeatures=50, nlevels=1, scale=1.2, min_length=0.02),
               dict(nfeatures=200, nlevels=3, scale=1.2, min_length=0.1, fit_err=1.0),
               dict(nfeatures=100, nlevels=4, scale=1.5, min_length=0.02, fit_err=2.5)):
        for im in (img, img[:200, :260].copy(), S.image("urban1_1241x376.pgm")):
            r, o = OracleLines(names, **kw), oracle.lines(**kw)
            rk, rd = r.extract(im)
            ok, od = o.extract(im)
            assert rk.tobytes() == ok.tobytes() and np.array_equal(rd, od), f"{kw} {im.shape}"
    flat = np.full((240, 320), 99, np.uint8)
    assert len(OracleLines(names).extract(flat)[0]) == 0 and len(oracle.lines().extract(flat)[0]) == 0


@needs_ref
@pytest.mark.parametrize("name", S.IMAGES[:4])
def test_lines_on_the_orb_pyramid_equal_the_reference(oracle, name):
    """Frame::PrecomputeGaussianPyramid run by the reference itself (the line extractor reads the ORB levels as regions
    of interest inside their bordered buffers) against the restatement's tightly packed hand-over."""
    img = S.image(name)
    rk, rd = ref_extractors()[2](img)
    ok, od = oracle_extractors(oracle)[2](img)
    assert len(rk) > 5 and rk.tobytes() == ok.tobytes() and np.array_equal(rd, od)


# ------------------------------------------------------------------ the committed digests
def _digests():
    with open(DIGESTS) as f:
        return json.load(f)["cases"]


@needs_ref
def test_committed_digests_are_what_the_compiled_reference_produces():
    assert S.run(*ref_extractors()) == _digests(), "re-run scripts/make_frontend_golden.py"


def test_oracle_reproduces_the_reference_digests(oracle):
    got, want = S.run(*oracle_extractors(oracle)), _digests()
    for name in S.IMAGES:
        for case, rec in want[name].items():
            assert got[name][case] == rec, f"{name} / {case}"


@pytest.mark.gpu
def test_hip_reproduces_the_reference_digests():
    """The product path (C ABI -> HIP kernels + host stages) against digests made by the reference's own source."""
    import torch
    from plvs_amd.lines import LineExtractor
    from plvs_amd.orb import ORBextractor

    keep = []

    def make_orb(nf):
        e = ORBextractor(nf, 1.2, 8, 20, 7)
        keep.append(e)
        return lambda img, lap: e(img, None, lap)

    def make_lines(nf):
        e = LineExtractor(nf)
        keep.append(e)
        return lambda img: e(img)

    def make_shared(img):
        orb, lines = ORBextractor(1000, 1.2, 8, 20, 7), LineExtractor(100)
        keep.extend([orb, lines])
        lines.SetGaussianPyramid(orb)
        orb(torch.from_numpy(np.ascontiguousarray(img)).cuda())
        return lines(img)

    got, want = S.run(make_orb, make_lines, make_shared), _digests()
    for name in S.IMAGES:
        for case, rec in want[name].items():
            assert got[name][case] == rec, f"{name} / {case}"
    for e in keep:
        e.close()
