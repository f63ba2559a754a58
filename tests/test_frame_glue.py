"""The per-frame glue between extraction and the searches: Frame::UndistortKeyPoints,
ComputeImageBounds, UndistortKeyLines, AssignFeaturesToGrid (reference src/Frame.cc:1507-1778, 716-746).

CPU: oracle/frame_glue.cpp against the reference's OWN Frame.cc (oracle/_ref/libmatchers_ref.so, compiled unmodified —
cv::undistortPoints and cv::fastAtan2 are the repository's restatements of the OpenCV primitives on both sides).
GPU: plvs_amd/csrc/frame_glue.hip through the C ABI against the oracle, bit for bit."""
import ctypes
import os

import numpy as np
import pytest

from tests.oracle_lib import KEYLINE_DTYPE, KP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libmatchers_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libmatchers_ref.so (built where /root/reference exists) not present")

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
TUM1_D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)       # Examples_old/RGB-D/TUM1.yaml
CALIBS = [(TUM1_K, TUM1_D), (TUM1_K, TUM1_D[:4].copy()),
          (np.array([458.654, 457.296, 367.215, 248.375], np.float32), np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)),
          (TUM1_K, np.zeros(5, np.float32)), (TUM1_K, None)]


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _keypoints(seed, n, w=640, h=480):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, KP_DTYPE)
    k["x"], k["y"] = rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)       # a few outside the image
    k["x"][:4], k["y"][:4] = [0, w, 0, w], [0, 0, h, h]
    k["octave"] = rng.integers(0, 8, n)
    k["angle"], k["size"], k["response"] = rng.uniform(0, 360, n), 31, rng.uniform(1, 100, n)
    return k


def _keylines(seed, n, w=640, h=480):
    rng = np.random.default_rng(seed)
    kl = np.zeros(n, KEYLINE_DTYPE)
    cx, cy, ang, ln = rng.uniform(0, w, n), rng.uniform(0, h, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(10, 200, n)
    kl["startPointX"], kl["startPointY"] = cx - 0.5 * ln * np.cos(ang), cy - 0.5 * ln * np.sin(ang)
    kl["endPointX"], kl["endPointY"] = cx + 0.5 * ln * np.cos(ang), cy + 0.5 * ln * np.sin(ang)
    kl["angle"], kl["octave"], kl["class_id"], kl["lineLength"] = ang, rng.integers(0, 3, n), np.arange(n), ln
    kl["response"], kl["numOfPixels"] = rng.uniform(0, 1, n), ln.astype(np.int32)
    return kl


class _Side:
    """The four functions of one implementation (prefix ref_frame_ / oracle_frame_)."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix

    def undistort_keypoints(self, k, K, D):
        un = np.empty_like(k)
        getattr(self.lib, self.p + "undistort_keypoints")(_p(k), len(k), _p(K), _p(D), 0 if D is None else len(D), _p(un))
        return un

    def bounds(self, w, h, K, D):
        b = np.zeros(5, np.float32)
        getattr(self.lib, self.p + "compute_image_bounds")(w, h, _p(K), _p(D), 0 if D is None else len(D), _p(b))
        return b

    def undistort_keylines(self, kl, K, D, b):
        un, kept = np.empty_like(kl), np.zeros(max(len(kl), 1), np.int32)
        f = getattr(self.lib, self.p + "undistort_keylines")
        f.restype = _i
        b4 = np.ascontiguousarray(b[:4], np.float32)
        m = f(_p(kl), len(kl), _p(K), _p(D), 0 if D is None else len(D), _p(b4), _p(un), _p(kept))
        return un[:m], kept[:m]

    def grid(self, un, mnx, mny, iw, ih):
        start, items = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(len(un), 1), np.int32)
        f = getattr(self.lib, self.p + "assign_features_to_grid")
        f.restype = _i
        f.argtypes = [_vp, _i, _f, _f, _f, _f, _vp, _vp]
        m = f(_p(un), len(un), mnx, mny, iw, ih, _p(start), _p(items))
        return start, items[:m]


class _Hip:
    def undistort_keypoints(self, k, K, D):
        from plvs_amd.frame import UndistortKeyPoints
        return UndistortKeyPoints(k, K, D)

    def bounds(self, w, h, K, D):
        from plvs_amd.frame import ComputeImageBounds
        return np.array(ComputeImageBounds(w, h, K, D), np.float32)

    def undistort_keylines(self, kl, K, D, b):
        from plvs_amd.frame import UndistortKeyLines
        return UndistortKeyLines(kl, K, D, b)

    def grid(self, un, mnx, mny, iw, ih):
        from plvs_amd.frame import AssignFeaturesToGrid
        return AssignFeaturesToGrid(un, mnx, mny, iw, ih)


def _same(a, b):
    assert a.tobytes() == b.tobytes()


def _check(want, got, seed):
    for ci, (K, D) in enumerate(CALIBS):
        k = _keypoints(seed + ci, 2100 if ci == 0 else 300)
        wu, gu = want.undistort_keypoints(k, K, D), got.undistort_keypoints(k, K, D)
        _same(wu, gu)
        distorted = D is not None and D[0] != 0
        assert (wu["x"] != k["x"]).any() == distorted
        wb, gb = want.bounds(640, 480, K, D), got.bounds(640, 480, K, D)
        _same(wb, gb)
        assert wb[1] - wb[0] > 550 and wb[4] > 700
        iw, ih = np.float32(64.0) / np.float32(wb[1] - wb[0]), np.float32(48.0) / np.float32(wb[3] - wb[2])   # Frame.cc:448-449
        ws, wi = want.grid(wu, float(wb[0]), float(wb[2]), float(iw), float(ih))
        gs, gi = got.grid(gu, float(wb[0]), float(wb[2]), float(iw), float(ih))
        _same(ws, gs)
        _same(wi, gi)
        assert 0.8 * len(k) < len(wi) <= len(k)
        kl = _keylines(seed + 50 + ci, 160)
        (wl, wk), (gl, gk) = want.undistort_keylines(kl, K, D, wb), got.undistort_keylines(kl, K, D, wb)
        _same(wk, gk)
        _same(np.ascontiguousarray(wl), np.ascontiguousarray(gl))
        assert 0 < len(wk) <= len(kl) and (len(wk) < len(kl)) == distorted
    # empty inputs
    K, D = CALIBS[0]
    assert len(got.undistort_keypoints(np.zeros(0, KP_DTYPE), K, D)) == 0
    s0, i0 = got.grid(np.zeros(0, KP_DTYPE), 0.0, 0.0, 0.1, 0.1)
    assert len(i0) == 0 and (s0 == 0).all()
    l0, k0 = got.undistort_keylines(np.zeros(0, KEYLINE_DTYPE), K, D, np.array([0, 640, 0, 480], np.float32))
    assert len(l0) == 0 and len(k0) == 0


@needs_ref
def test_oracle_glue_equals_the_reference_source(oracle):
    _check(_Side(ctypes.CDLL(REF), "ref_frame_"), _Side(oracle.lib, "oracle_frame_"), 1)


@pytest.mark.gpu
def test_hip_glue_equals_the_oracle(oracle):
    _check(_Side(oracle.lib, "oracle_frame_"), _Hip(), 7)


def _distort(xy_un, K, D):
    """The forward Brown model in float64 (OpenCV's projectPoints with R = I, t = 0): undistorted pixels -> distorted pixels."""
    fx, fy, cx, cy = [float(v) for v in K]
    k1, k2, p1, p2 = [float(v) for v in D[:4]]
    k3 = float(D[4]) if len(D) > 4 else 0.0
    x, y = (xy_un[:, 0] - cx) / fx, (xy_un[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], -1)


def _round_trip(side):
    """An analytic check that rests on nothing restated: distort known pixels with the forward model, undistort
    them — the fixed-point iteration of cv::undistortPoints (five rounds) must come back to where they started.  How close
    depends on the distortion at the pixel (five rounds are OpenCV's count, not enough to converge in the corners of a
    strongly distorted image): TUM1 under 0.05 px everywhere, EuRoC (k1 = -0.28) under 0.05 px for 9 pixels in 10 and
    0.23 px in the corners, exact without distortion."""
    rng = np.random.default_rng(4)
    for K, D, tol, tol90 in ((TUM1_K, TUM1_D, 0.05, 0.05), (CALIBS[2][0], CALIBS[2][1], 0.3, 0.05),
                             (TUM1_K, np.zeros(5, np.float32), 2e-4, 2e-4)):
        un = np.stack([rng.uniform(20, 620, 500), rng.uniform(20, 460, 500)], -1)
        d = _distort(un, K, D)
        k = np.zeros(len(d), KP_DTYPE)
        k["x"], k["y"] = d[:, 0], d[:, 1]
        got = side.undistort_keypoints(k, K, D)
        # (the input itself is rounded to float32 pixels: 3e-5 px at 640)
        err = np.hypot(got["x"].astype(np.float64) - un[:, 0], got["y"].astype(np.float64) - un[:, 1])
        assert err.max() < tol and np.quantile(err, 0.9) < tol90, (float(err.max()), float(np.quantile(err, 0.9)), K, D)
        assert np.array_equal(got["octave"], k["octave"]) and np.array_equal(got["angle"], k["angle"])


def test_oracle_undistort_inverts_the_forward_model(oracle):
    _round_trip(_Side(oracle.lib, "oracle_frame_"))


@pytest.mark.gpu
def test_hip_undistort_inverts_the_forward_model():
    _round_trip(_Hip())
