"""Line front end (EDLines + LBD): oracle properties and the product's host
stages vs the oracle (CPU); full HIP + host path vs oracle, stage by stage (GPU).

Bar: bit-exact KeyLine fields (compared as raw 68-byte records) and descriptor bits.
"""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib
from tests.oracle_lib import golden
from tests.test_orb import synth_frame

IMAGES = ["aloe_640x480.pgm", "cones_640x480.pgm", "urban1_1241x376.pgm"]


def test_oracle_lines_properties(oracle):
    e = oracle.lines()
    img = golden("aloe_640x480.pgm")
    kl, d = e.extract(img)
    assert len(kl) == 100 and d.shape == (100, 32)
    assert np.all(np.diff(kl["response"]) <= 0)                    # sorted by response
    assert list(kl["class_id"]) == list(range(100))
    assert np.all(kl["response"] >= 0.02)
    assert set(np.unique(kl["octave"])) <= {0, 1, 2}
    assert np.all(kl["numOfPixels"] >= 15)
    L = np.hypot(kl["ePointInOctaveX"] - kl["sPointInOctaveX"], kl["ePointInOctaveY"] - kl["sPointInOctaveY"])
    assert np.allclose(L * np.float32(1.2) ** kl["octave"], kl["lineLength"], rtol=1e-4)
    # direction agrees with the endpoint order chosen by OctaveKeyLines
    ang = np.arctan2(kl["ePointInOctaveY"] - kl["sPointInOctaveY"], kl["ePointInOctaveX"] - kl["sPointInOctaveX"])
    dd = np.abs(np.angle(np.exp(1j * (ang - kl["angle"]))))
    assert np.all(dd < 0.2)
    flat = np.full((240, 320), 99, np.uint8)
    k2, d2 = e.extract(flat)
    assert len(k2) == 0


def test_oracle_line_maps(oracle):
    """Sobel / gradient maps are exact integer functions of the blurred image."""
    e = oracle.lines()
    e.extract(golden("cones_640x480.pgm"))
    for o in range(3):
        b = e.octave_map(o, "blur").astype(np.int32)
        p = np.pad(b, 1, mode="reflect")
        dx = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
        dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
        assert np.array_equal(e.octave_map(o, "dx"), dx)
        assert np.array_equal(e.octave_map(o, "dy"), dy)
        s = np.abs(dx) + np.abs(dy)
        g = np.rint(np.where(s > 81, s, 0) * 0.25).astype(np.int16)   # numpy rint = round half even
        assert np.array_equal(e.octave_map(o, "g"), g)
        assert np.array_equal(e.octave_map(o, "dir"), np.where(np.abs(dx) < np.abs(dy), 255, 0))


def _run_hostlines(lib, ora, img, nfeatures=100):
    sizes, gds, dxs, dys = [], [], [], []
    for o in range(3):
        g = ora.octave_map(o, "g").astype(np.uint16)
        d = ora.octave_map(o, "dir")
        gds.append(np.ascontiguousarray(g | np.where(d == 255, 0x8000, 0).astype(np.uint16)))
        dxs.append(np.ascontiguousarray(ora.octave_map(o, "dx")))
        dys.append(np.ascontiguousarray(ora.octave_map(o, "dy")))
        sizes += [g.shape[1], g.shape[0]]
    sizes = np.array(sizes, np.int32)
    P = ctypes.c_void_p
    arr = lambda xs: (P * 3)(*[x.ctypes.data_as(P) for x in xs])
    out = np.zeros(20000, oracle_lib.KEYLINE_DTYPE)
    per = np.zeros(3, np.int32)
    lib.hostlines_run.argtypes = [ctypes.c_int, P, P, P, P, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_double, ctypes.c_double, P, ctypes.c_int, P]
    n = lib.hostlines_run(3, sizes.ctypes.data_as(P), arr(gds), arr(dxs), arr(dys), 1.2, nfeatures, img.shape[1],
                          img.shape[0], 0.02, 1.6, out.ctypes.data_as(P), len(out), per.ctypes.data_as(P))
    return out[:n], per


@pytest.mark.parametrize("name", IMAGES + ["synth3"])
def test_product_host_stages_match_oracle(oracle, name):
    """lines_host.hpp (linking, fitting, validation, grouping, selection), fed with the
    oracle's per-pixel maps, reproduces the oracle's KeyLines byte for byte."""
    lib = oracle_lib.load_hostlines()
    img = synth_frame(3) if name == "synth3" else golden(name)
    ora = oracle.lines()
    okl, _ = ora.extract(img)
    kl, per = _run_hostlines(lib, ora, img)
    assert list(per) == [ora.num_in_octave(o) for o in range(3)]
    assert len(kl) == len(okl) > 10
    assert kl.tobytes() == okl.tobytes()
    # the same with the anchor test taken from a flag map, as the device hands it over
    os.environ["PLVS_HOSTLINES_FLAGS"] = "1"
    try:
        kl_f, per_f = _run_hostlines(lib, ora, img)
    finally:
        del os.environ["PLVS_HOSTLINES_FLAGS"]
    assert list(per_f) == list(per) and kl_f.tobytes() == okl.tobytes()
    # nfeatures = 0: no sort, so the min-length cut falls at the first short line in
    # detection order (the reference's behaviour, src/LineExtractor.cc:229-266)
    ora_all = oracle.lines(nfeatures=0)
    okl_all, _ = ora_all.extract(img)
    kl_all, _ = _run_hostlines(lib, ora_all, img, nfeatures=0)
    assert kl_all.tobytes() == okl_all.tobytes() and len(kl_all) > 0


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", IMAGES + ["synth3", "synth9"])
def test_hip_lines_match_oracle_stage_by_stage(oracle, name):
    from plvs_amd.lines import LineExtractor
    img = synth_frame(int(name[5:])) if name.startswith("synth") else golden(name)
    dev = LineExtractor(100)
    ora = oracle.lines()
    okl, odesc = ora.extract(img)
    kl, desc = dev(img)
    for o in range(3):
        assert np.array_equal(dev.octave_map(o, "blur"), ora.octave_map(o, "blur")), f"blur octave {o}"
        assert np.array_equal(dev.octave_map(o, "dx"), ora.octave_map(o, "dx")), f"dx octave {o}"
        assert np.array_equal(dev.octave_map(o, "dy"), ora.octave_map(o, "dy")), f"dy octave {o}"
        gd = dev.octave_map(o, "gd")
        assert np.array_equal((gd & 0x1ff).astype(np.int16), ora.octave_map(o, "g")), f"gradient octave {o}"
        assert np.array_equal(np.where(gd & 0x8000, 255, 0), ora.octave_map(o, "dir")), f"direction octave {o}"
        assert dev.num_in_octave(o) == ora.num_in_octave(o), f"segments in octave {o}"
    assert len(kl) == len(okl)
    assert kl.tobytes() == okl.tobytes(), "KeyLine records differ"
    assert np.array_equal(desc, odesc), "LBD descriptor bits differ"
    dev.close()


@pytest.mark.gpu
def test_hip_lines_device_input_resize_and_empty(oracle):
    import torch
    from plvs_amd.lines import LineExtractor, LSDOptions
    dev = LineExtractor(0, LSDOptions(numOctaves=2, scale=1.4142135, min_length=0.05))
    ora = oracle.lines(nfeatures=0, nlevels=2, scale=1.4142135, min_length=0.05)
    img = golden("cones_640x480.pgm")
    okl, odesc = ora.extract(img)
    kl, desc = dev(torch.from_numpy(img).cuda())
    assert kl.tobytes() == okl.tobytes() and np.array_equal(desc, odesc) and len(kl) > 0
    small = img[:200, :260].copy()                     # new geometry on the same handle
    o2 = ora.extract(small)
    d2 = dev(small)
    assert d2[0].tobytes() == o2[0].tobytes() and np.array_equal(d2[1], o2[1])
    flat = np.full((240, 320), 99, np.uint8)
    kl0, d0 = dev(flat)
    assert len(kl0) == 0 and d0.shape == (0, 32)
    dev.close()


@pytest.mark.gpu
def test_hip_frame_extract_points_and_lines_together(oracle):
    """Frame.cc:503-508: ORB and line extraction of one image on two threads — the
    combined entry point returns exactly what the two extractors return alone."""
    import torch
    from plvs_amd.frame import extract_frame
    from plvs_amd.lines import LineExtractor
    from plvs_amd.orb import ORBextractor
    orb, lines = ORBextractor(1000, 1.2, 8, 20, 7), LineExtractor(100)
    oorb, olines = oracle.orb(1000, 1.2, 8, 20, 7), oracle.lines()
    for name in IMAGES[:2]:
        img = golden(name)
        for _ in range(3):   # repeated calls: the overlapped host threads must not disturb the order
            mono, kps, desc, kl, ldesc = extract_frame(orb, lines, torch.from_numpy(img).cuda())
            omono, okps, odesc = oorb.extract(img)
            okl, oldesc = olines.extract(img)
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
            assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
    orb.close()
    lines.close()


@pytest.mark.gpu
def test_hip_frame_extract_hook_sees_the_points_before_the_lines_are_out(oracle):
    """plvs_hip_frame_extract_dev_hook: the hook runs once, on the caller's thread, with the finished points — the same
    points and lines come back as without it; an exception raised inside it reaches the caller after the call."""
    import threading
    import torch
    from plvs_amd.frame import extract_frame
    from plvs_amd.lines import LineExtractor
    from plvs_amd.orb import ORBextractor
    orb, lines = ORBextractor(1000, 1.2, 8, 20, 7), LineExtractor(100)
    img = golden(IMAGES[0])
    omono, okps, odesc = oracle.orb(1000, 1.2, 8, 20, 7).extract(img)
    okl, oldesc = oracle.lines().extract(img)
    seen = []

    def hook(kps, desc):
        seen.append((kps.tobytes(), desc.copy(), threading.get_ident()))
        return len(kps)

    for _ in range(3):
        mono, kps, desc, kl, ldesc, hooked = extract_frame(orb, lines, torch.from_numpy(img).cuda(), after_points=hook)
        assert hooked == len(okps)
        assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
        assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
    assert len(seen) == 3
    for b, d, tid in seen:
        assert b == okps.tobytes() and np.array_equal(d, odesc) and tid == threading.get_ident()

    def bad(kps, desc):
        raise ValueError("from the hook")

    with pytest.raises(ValueError, match="from the hook"):
        extract_frame(orb, lines, torch.from_numpy(img).cuda(), after_points=bad)
    mono, kps, desc, kl, ldesc = extract_frame(orb, lines, torch.from_numpy(img).cuda())   # the extractors are intact
    assert kps.tobytes() == okps.tobytes() and kl.tobytes() == okl.tobytes()
    orb.close()
    lines.close()


# ------------------------------------------------------------------ shared ORB pyramid (O9)
def _oracle_shared(oracle, img, nfeatures=1000):
    oorb = oracle.orb(nfeatures, 1.2, 8, 20, 7)
    out = oorb.extract(img)
    ol = oracle.lines()
    ol.set_pyramid([oorb.level(l) for l in range(8)], 3, 1.2)
    return out, ol


def test_oracle_lines_on_a_shared_pyramid(oracle):
    """Line.pyramidPrecomputation: octave i = ORB level i (cvRound sizes, unblurred), no resize chain."""
    img = golden(IMAGES[0])
    _, ol = _oracle_shared(oracle, img)
    kl, desc = ol.extract(img)
    own_kl, _ = oracle.lines().extract(img)
    assert 20 < len(kl) <= 100 and desc.shape == (len(kl), 32)
    assert ol.octave_size(0) == (640, 480) and ol.octave_size(1) == (533, 400) and ol.octave_size(2) == (444, 333)
    # octave 0 is the same image either way; the higher octaves differ (resized from the unblurred level)
    k0, o0 = kl[kl["octave"] == 0], own_kl[own_kl["octave"] == 0]
    assert len(k0) > 5 and len(o0) > 5
    assert kl.tobytes() != own_kl.tobytes()


@pytest.mark.gpu
def test_hip_lines_on_a_shared_pyramid_match_oracle(oracle):
    import torch
    from plvs_amd.frame import extract_frame
    from plvs_amd.lines import LineExtractor
    from plvs_amd.orb import ORBextractor
    orb, lines = ORBextractor(1000, 1.2, 8, 20, 7), LineExtractor(100)
    lines.SetGaussianPyramid(orb)
    for name in IMAGES[:3]:
        img = golden(name)
        (omono, okps, odesc), ol = _oracle_shared(oracle, img)
        okl, oldesc = ol.extract(img)
        # sequential: extractor first, then the lines on its pyramid
        orb(img)
        kl, ldesc = lines(img)
        assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
        for o in range(3):
            assert np.array_equal(lines.octave_map(o, "blur"), ol.octave_map(o, "blur"))
        # concurrent: the frame-level entry point orders the two through the pyramid event
        for _ in range(3):
            mono, kps, desc, kl, ldesc = extract_frame(orb, lines, torch.from_numpy(img).cuda())
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
            assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
    # a different image size rebuilds both geometries
    small = golden(IMAGES[0])[:300, :400].copy()
    (_, _, _), ol = _oracle_shared(oracle, small)
    okl, oldesc = ol.extract(small)
    mono, kps, desc, kl, ldesc = extract_frame(orb, lines, torch.from_numpy(small).cuda())
    assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
    # back to the extractor's own pyramid
    lines.SetGaussianPyramid(None)
    img = golden(IMAGES[0])
    okl, oldesc = oracle.lines().extract(img)
    kl, ldesc = lines(img)
    assert kl.tobytes() == okl.tobytes() and np.array_equal(ldesc, oldesc)
    # sharing with an extractor that has not seen an image is an error, not a fallback
    from plvs_amd import _lib
    fresh = ORBextractor(500, 1.2, 8, 20, 7)
    lines.SetGaussianPyramid(fresh)
    with pytest.raises(_lib.PlvsHipError):
        lines(img)
    lines.close()
    orb.close()
