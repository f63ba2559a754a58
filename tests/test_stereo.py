"""Sparse stereo matching (Frame::ComputeStereoMatches): the oracle's properties on
CPU, and the HIP path against the oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.oracle_lib import golden

KITTI_FX, KITTI_BF = 718.856, 386.1448           # Examples_old/Stereo/KITTI00-02.yaml
MB = np.float32(KITTI_BF / KITTI_FX)             # Frame.cc: mb = mbf / fx
NLEVELS, SCALE = 8, 1.2


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def scale_tables(nlevels=NLEVELS, factor=SCALE):
    """ORBextractor.cc:455-470: float recurrences."""
    s = [np.float32(1.0)]
    for _ in range(1, nlevels):
        s.append(np.float32(s[-1] * np.float32(factor)))
    s = np.array(s, np.float32)
    return s, (np.float32(1.0) / s).astype(np.float32)


def pair(name):
    if name == "urban1":
        return golden("urban1_1241x376.pgm"), golden("urban1_right_1241x376.pgm")
    if name == "shift17":                            # constant disparity 17 px
        img = golden("aloe_640x480.pgm")
        return np.ascontiguousarray(img[:, :-17]), np.ascontiguousarray(img[:, 17:])
    if name == "swapped":                            # negative disparities: (almost) nothing survives
        return golden("urban1_right_1241x376.pgm"), golden("urban1_1241x376.pgm")
    raise KeyError(name)


def oracle_side(oracle, left, right, nfeatures):
    out = []
    for img in (left, right):
        ex = oracle.orb(nfeatures, SCALE, NLEVELS, 20, 7)
        _, k, d = ex.extract(img)
        out.append((k, d, [ex.level(l) for l in range(NLEVELS)]))
    return out


def test_oracle_recovers_a_known_disparity(oracle):
    left, right = pair("shift17")
    (kl, dl, pl), (kr, dr, pr) = oracle_side(oracle, left, right, 1000)
    s, inv = scale_tables()
    u, z, score, kept = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, s, inv, MB, np.float32(KITTI_BF))
    ok = u >= 0
    assert kept == int(ok.sum()) and kept > 0.5 * kl.shape[0]
    disp = kl["x"][ok] - u[ok]
    assert np.abs(disp - 17.0).max() < 1.5 * s[kl["octave"][ok]].max()
    assert np.median(np.abs(disp - 17.0)) < 0.2
    np.testing.assert_array_equal(z[ok], np.float32(KITTI_BF) / disp.astype(np.float32))
    assert (z[~ok] == -1).all() and (score[ok] >= 0).all()


def test_oracle_on_a_real_pair(oracle):
    left, right = pair("urban1")
    (kl, dl, pl), (kr, dr, pr) = oracle_side(oracle, left, right, 2000)
    s, inv = scale_tables()
    u, z, score, kept = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, s, inv, MB, np.float32(KITTI_BF))
    ok = u >= 0
    assert kept == int(ok.sum()) and kept > 0.25 * kl.shape[0]
    disp = kl["x"][ok] - u[ok]
    assert (disp > 0).all() and (disp < KITTI_FX).all() and (z[ok] > 0).all()
    # the median cut: every survivor's score is below 1.5 * 1.4 * median of the pre-cut scores
    pre = np.sort(score[score >= 0])
    th = np.float32(1.5) * np.float32(1.4) * np.float32(pre[pre.shape[0] // 2])
    assert (score[ok] < th).all() and (score[(score >= 0) & ~ok] >= th).all()
    # nothing on the right -> nothing matched, and no crash on the empty median
    u0, z0, _, kept0 = oracle.stereo_matches(kl, dl, kr[:0], dr[:0], pl, pr, s, inv, MB, np.float32(KITTI_BF))
    assert kept0 == 0 and (u0 == -1).all() and (z0 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,nfeatures", [("urban1", 2000), ("shift17", 1000), ("swapped", 1000)])
def test_hip_stereo_matches_oracle(oracle, name, nfeatures):
    from plvs_amd.orb import ORBextractor
    from plvs_amd.stereo import StereoMatcher
    left, right = pair(name)
    (kl, dl, pl), (kr, dr, pr) = oracle_side(oracle, left, right, nfeatures)
    exl, exr = ORBextractor(nfeatures, SCALE, NLEVELS, 20, 7), ORBextractor(nfeatures, SCALE, NLEVELS, 20, 7)
    _, hkl, hdl = exl(left)
    _, hkr, hdr = exr(right)
    assert hkl.tobytes() == kl.tobytes() and hkr.tobytes() == kr.tobytes()       # same front end
    assert hdl.tobytes() == dl.tobytes() and hdr.tobytes() == dr.tobytes()
    s, inv = scale_tables()
    np.testing.assert_array_equal(np.asarray(exl.GetScaleFactors(), np.float32), s)
    sm = StereoMatcher(exl, exr)
    want_u, want_z, _, kept = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, s, inv, MB, np.float32(KITTI_BF))
    got_u, got_z = sm.ComputeStereoMatches(hkl, hdl, hkr, hdr, MB, np.float32(KITTI_BF))
    assert got_u.tobytes() == want_u.tobytes()
    assert got_z.tobytes() == want_z.tobytes()
    if name != "swapped":
        assert kept > 0.25 * kl.shape[0]
    # degenerate inputs
    u0, z0 = sm.ComputeStereoMatches(hkl, hdl, hkr[:0], hdr[:0], MB, np.float32(KITTI_BF))
    assert (u0 == -1).all() and (z0 == -1).all()
    u1, z1 = sm.ComputeStereoMatches(hkl[:0], hdl[:0], hkr, hdr, MB, np.float32(KITTI_BF))
    assert u1.shape == (0,) and z1.shape == (0,)
    # a subset of the left keypoints gives the same per-keypoint candidates, but its own median
    sub = slice(0, kl.shape[0] // 3)
    w_u, w_z, _, _ = oracle.stereo_matches(kl[sub], dl[sub], kr, dr, pl, pr, s, inv, MB, np.float32(KITTI_BF))
    g_u, g_z = sm.ComputeStereoMatches(hkl[sub], hdl[sub], hkr, hdr, MB, np.float32(KITTI_BF))
    assert g_u.tobytes() == w_u.tobytes() and g_z.tobytes() == w_z.tobytes()


@pytest.mark.gpu
def test_hip_stereo_needs_both_pyramids():
    from plvs_amd import _lib
    from plvs_amd.orb import ORBextractor, KP_DTYPE
    from plvs_amd.stereo import StereoMatcher
    sm = StereoMatcher(ORBextractor(500, SCALE, NLEVELS, 20, 7), ORBextractor(500, SCALE, NLEVELS, 20, 7))
    k = np.zeros(4, KP_DTYPE)
    with pytest.raises(_lib.PlvsHipError):
        sm.ComputeStereoMatches(k, np.zeros((4, 32), np.uint8), k, np.zeros((4, 32), np.uint8), 0.5, 380.0)
