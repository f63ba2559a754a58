"""The search oracles pinned by the reference's OWN sources: oracle/_ref/libmatchers_ref.so is src/ORBmatcher.cc,
src/LineMatcher.cc and src/Frame.cc (with ORBextractor.cc, LineExtractor.cc, the line_descriptor sources and DBoW2's
vector classes) compiled UNMODIFIED from /root/reference against force-included stand-ins for the classes that cannot
compile here (oracle/ref/slam_shim/: KeyFrame, MapPoint, MapLine, Tracking, IMU, Sophus, the camera models — data
holders; the arithmetic stays in the reference's sources).  oracle/orb_search.c, line_search.c, line_proj_search.c and
stereo.c must give the reference's assignments on the cases the GPU tests run the device against.

Where the product's interface takes a quantity "handed over by the caller" — the projection of a map point into the
current frame — the reference derives it inside the function from poses and world points: the test gives the reference
the poses and points and repeats its two lines of float arithmetic (Tcw * x3Dw, 1.0 / z, the pinhole projection) to
hand the SAME numbers to the oracle.

The .so is built where the reference tree exists (oracle/ref/Makefile, __graft_entry__.build()) and travels with the
snapshot; without it the tests are skipped (nothing here reads /root/reference at run time)."""
import ctypes
import os

import numpy as np
import pytest

from tests import test_orb_search as tos

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libmatchers_ref.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF),
                                reason="oracle/_ref/libmatchers_ref.so (built where /root/reference exists) not present")

_vp, _f, _i = ctypes.c_void_p, ctypes.c_float, ctypes.c_int


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


@pytest.fixture(scope="module")
def ref():
    return ctypes.CDLL(REF)


# ---------------------------------------------------------------- ORBmatcher::SearchByProjection(F, MapPoints)
@pytest.mark.parametrize("seed,th,far", [(1, 1.0, False), (2, 3.0, False), (5, 1.0, True), (8, 5.0, True), (3, 1.0, False)])
def test_search_by_projection_mappoints(oracle, ref, seed, th, far):
    F, M, occ = tos.make_case(seed)
    want_n, want = tos.oracle_search(oracle.lib, F, M, th, far, 40.0, 0.8, occ)
    fc, mc = F.as_c(), M.as_c()
    got = np.full(fc.n, -7, np.int32)
    fn = ref.ref_orb_search_by_projection
    fn.argtypes = [_vp, _vp, _f, _i, _f, _f, _vp, _vp]
    fn.restype = _i
    got_n = fn(ctypes.byref(fc), ctypes.byref(mc), th, int(far), 40.0, 0.8, _p(occ), _p(got))
    assert got_n == want_n > 50
    assert np.array_equal(got, want)


def test_search_by_projection_mappoints_small_and_empty(oracle, ref):
    fn = ref.ref_orb_search_by_projection
    fn.argtypes = [_vp, _vp, _f, _i, _f, _f, _vp, _vp]
    fn.restype = _i
    for n, m in ((40, 0), (1, 5), (300, 700)):
        F, M, occ = tos.make_case(11, n=n, m=m)
        want_n, want = tos.oracle_search(oracle.lib, F, M, 2.0, False, 0.0, 0.8, occ)
        fc, mc = F.as_c(), M.as_c()
        got = np.full(fc.n, -7, np.int32)
        assert fn(ctypes.byref(fc), ctypes.byref(mc), 2.0, 0, 0.0, 0.8, _p(occ), _p(got)) == want_n
        assert np.array_equal(got, want)


# ---------------------------------------------------------------- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame)
CAM = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)   # TUM1.yaml


def _pose(rng, angle_deg, t):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(angle_deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    return np.concatenate([R, np.asarray(t, float)[:, None]], axis=1).astype(np.float32)


def _transform(T, p):
    """Sophus::SE3f * point as the stand-in (and Eigen's 3x3 lazy product) evaluates it: row . p = a0 p0 + (a1 p1 + a2 p2), + t."""
    f = np.float32
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = (T[r, 0] * p[:, 0] + (T[r, 1] * p[:, 1] + T[r, 2] * p[:, 2])).astype(f) + T[r, 3]
    return out.astype(f)


def _handed_over(Tcw, xyz_w):
    """src/ORBmatcher.cc:1807-1817: x3Dc = Tcw * x3Dw; invzc = 1.0 / x3Dc(2); uv = mpCamera->project(x3Dc) (Pinhole)."""
    c = _transform(Tcw, xyz_w)
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (1.0 / c[:, 2].astype(np.float64)).astype(np.float32)
        u = (CAM[0] * c[:, 0] / c[:, 2] + CAM[2]).astype(np.float32)
        v = (CAM[1] * c[:, 1] / c[:, 2] + CAM[3]).astype(np.float32)
    return u, v, invz


@pytest.mark.parametrize("seed,th,direction,check", [(1, 15.0, 0, 1), (2, 7.0, 0, 1), (3, 15.0, 1, 1), (4, 30.0, 2, 0),
                                                     (6, 15.0, 0, 0)])
def test_search_by_projection_last_frame(oracle, ref, seed, th, direction, check):
    from plvs_amd.orbmatcher import LastFrameView
    F, cur_angle, max_x, max_y, mbf, L, occ = tos.make_ff_case(seed)
    rng = np.random.default_rng(seed + 500)
    Tcw = _pose(rng, 7.0, rng.uniform(-0.3, 0.3, 3))
    # world points whose projection is about what the case asks for (points behind the camera for invz < 0)
    with np.errstate(divide="ignore"):
        z = np.where(L.invz != 0, 1.0 / L.invz.astype(np.float64), 1e6)
    xc = (L.u.astype(np.float64) - CAM[2]) / CAM[0] * z
    yc = (L.v.astype(np.float64) - CAM[3]) / CAM[1] * z
    Rcw, tcw = Tcw[:, :3].astype(np.float64), Tcw[:, 3].astype(np.float64)
    xyz_w = ((np.stack([xc, yc, z], 1) - tcw) @ Rcw).astype(np.float32)          # R^T (x - t)
    u, v, invz = _handed_over(Tcw, xyz_w)
    L2 = LastFrameView(valid=L.valid, u=u, v=v, invz=invz, octave=L.octave, angle=L.angle, desc=L.desc, has_obs=L.has_obs)
    # the last frame's pose: tlc = Tlw * twc decides bForward / bBackward against mb (:1795-1796)
    mb = float(mbf / CAM[0])
    twc = -(Rcw.T @ tcw)
    dz = (0.0, 1.0, -1.0)[direction]
    Tlw = np.concatenate([np.eye(3), (-twc + np.array([0, 0, dz]))[:, None]], axis=1).astype(np.float32)
    want_n, want = tos.oracle_search_ff(oracle.lib, F, cur_angle, max_x, max_y, mbf, L2, th, int(direction == 1),
                                        int(direction == 2), check, occ)
    fc = F.as_c()
    got = np.full(fc.n, -7, np.int32)
    fn = ref.ref_orb_search_by_projection_ff
    fn.argtypes = [_vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _f, _i, _vp, _vp]
    fn.restype = _i
    c32 = lambda a, t: np.ascontiguousarray(a, t)
    valid, octave, angle = c32(L.valid, np.uint8), c32(L.octave, np.int32), c32(L.angle, np.float32)
    desc, has_obs = c32(L.desc, np.uint8), c32(L.has_obs, np.uint8)
    got_n = fn(ctypes.byref(fc), _p(c32(cur_angle, np.float32)), max_x, max_y, mbf, mb, _p(Tcw), _p(Tlw), _p(CAM), len(valid),
               _p(valid), _p(xyz_w), _p(octave), _p(angle), _p(desc), _p(has_obs), th, 0, 0.9, check, _p(occ), _p(got))
    assert got_n == want_n > 30
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- ORBmatcher::SearchByBoW(pKF, F)
@pytest.mark.parametrize("seed,ratio,check", [(1, 0.7, 1), (2, 0.7, 0), (3, 0.9, 1), (4, 0.5, 1)])
def test_search_by_bow(oracle, ref, seed, ratio, check):
    KV, kd, kv, ka, FV, fd, fa = tos.make_bow_case(seed)
    want_n, want = tos.oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, ratio, check)
    kc, fc = KV.as_c(), FV.as_c()
    got = np.full(fd.shape[0], -7, np.int32)
    fn = ref.ref_orb_search_by_bow
    fn.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _vp]
    fn.restype = _i
    got_n = fn(ctypes.byref(kc), _p(kd), kd.shape[0], _p(kv), _p(ka), ctypes.byref(fc), _p(fd), fd.shape[0], _p(fa), ratio,
               check, _p(got))
    assert got_n == want_n > 100
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- LineMatcher::SearchByKnn (F-F, KF-F), SearchStereoMatchesByKnn
from tests import test_line_search as tls  # noqa: E402


def _knn_fn(ref, name):
    fn = getattr(ref, name)
    fn.restype = _i
    fn.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _f, _i, _vp]
    return fn


@pytest.mark.parametrize("seed,ratio,check", [(1, 0.8, True), (2, 0.7, True), (3, 0.9, False), (4, 0.8, True), (7, 0.6, True)])
def test_lines_search_by_knn_last_frame(oracle, ref, seed, ratio, check):
    case = tls.make_case(seed, n_last=90 + 7 * seed, n_cur=100 + 3 * seed, rot=0.3 * seed)
    want_n, want = tls.run(tls.oracle_fn(oracle), case, ratio, check)
    got_n, got = tls.run(_knn_fn(ref, "ref_lines_search_by_knn"), case, ratio, check)
    assert got_n == want_n > 10
    assert np.array_equal(got, want)
    # a single current line: the k-NN has no second neighbour
    last, valid, ang_last, cur, ang_cur = case
    one = (last, valid, ang_last, cur[:1].copy(), ang_cur[:1].copy())
    w = tls.run(tls.oracle_fn(oracle), one, ratio, check)
    g = tls.run(_knn_fn(ref, "ref_lines_search_by_knn"), one, ratio, check)
    assert g[0] == w[0] and np.array_equal(g[1], w[1])


@pytest.mark.parametrize("seed,ratio,check", [(1, 0.8, True), (2, 0.7, True), (3, 0.9, False), (5, 0.8, True)])
def test_lines_search_by_knn_key_frame(oracle, ref, seed, ratio, check):
    case = tls.make_case(10 + seed, n_last=120 + 7 * seed, n_cur=100 + 3 * seed, rot=0.4 * seed)
    want_n, want = tls.run(tls.oracle_kf_fn(oracle), case, ratio, check)
    got_n, got = tls.run(_knn_fn(ref, "ref_lines_search_by_knn_kf"), case, ratio, check)
    assert got_n == want_n > 10
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed,ratio,check,dd", [(1, 0.8, True, 50), (2, 0.7, True, 60), (3, 0.9, False, 50), (4, 0.8, True, 256)])
def test_lines_search_stereo_by_knn(oracle, ref, seed, ratio, check, dd):
    case = tls.make_stereo_case(seed, n_left=100 + 9 * seed, n_right=90 + 5 * seed)
    want_n, wq, wt, wd, wv = tls.oracle_stereo(oracle, case, ratio, check, dd)
    left, ang_l, oct_l, right, ang_r, oct_r = case
    cap = max(right.shape[0], 1)
    mq, mt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    md, mv = np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
    n_out = ctypes.c_int()
    fn = ref.ref_lines_search_stereo_by_knn
    fn.restype = _i
    fn.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]
    got_n = fn(_p(left), left.shape[0], _p(ang_l), _p(oct_l), _p(right), right.shape[0], _p(ang_r), _p(oct_r), ratio, int(check),
               dd, _p(mq), _p(mt), _p(md), _p(mv), ctypes.byref(n_out))
    k = n_out.value
    assert got_n == want_n > 5 and k == len(wq)
    assert np.array_equal(mq[:k], wq) and np.array_equal(mt[:k], wt)
    assert np.array_equal(md[:k], wd) and np.array_equal(mv[:k].astype(bool), wv)


# ---------------------------------------------------------------- LineMatcher::SearchByProjection (F, MapLines) and (F, LastF)
from tests import test_line_proj_search as tlp  # noqa: E402


def _line_view(c):
    from plvs_amd.linematcher import line_frame_view
    return line_frame_view(c["kl"], c["desc"], tlp.SCALE, tlp.INV_SIGMA2, tlp.MAX_DIAG, u_right_start=c["urs"],
                           u_right_end=c["ure"], bf=c["bf"])


@pytest.mark.parametrize("seed,stereo,edge,larger,ratio", [(1, False, False, False, 0.8), (2, True, False, False, 0.8),
                                                          (3, False, True, True, 0.9), (4, True, True, False, 0.7),
                                                          (5, True, False, True, 0.8)])
def test_lines_search_by_projection_maplines(oracle, ref, seed, stereo, edge, larger, ratio):
    c = tlp.make_case(seed, n_cur=150, n_last=130, stereo=stereo, theta_edge=edge)
    want_n, want = tlp.oracle_map(oracle, c, larger, ratio)
    F, keep = _line_view(c)
    got = np.full(len(c["kl"]), -7, np.int32)
    fn = ref.ref_lines_search_by_projection
    fn.restype = _i
    fn.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]
    got_n = fn(ctypes.byref(F), tlp._p(c["occupied"]), len(c["valid"]), tlp._p(c["valid"]), tlp._p(c["proj_map"]),
               tlp._p(c["octave"]), tlp._p(c["ldesc"]), tlp._p(c["has_obs"]), int(larger), ratio, _p(got))
    assert got_n == want_n > 20
    assert np.array_equal(got, want)


def _f32(a):
    return np.asarray(a, np.float32)


def _project_lines(Tcw, xyz_w, bounds):
    """LineProjection::ProjectLineWithCheck (include/LineProjection.h:145-262) for a pinhole camera, in the reference's
    float arithmetic: middle point in front and inside, end points in front and inside; invSz = 1.0f / z; the distance
    test passes (the stand-in map lines accept every distance).  -> ok [n], proj [n, 6]."""
    S, E = xyz_w[:, :3], xyz_w[:, 3:]
    M = (np.float32(0.5) * (S + E)).astype(np.float32)
    ok = np.ones(len(S), bool)
    out = np.zeros((len(S), 6), np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        for P, col in ((M, None), (S, 0), (E, 2)):
            c = _transform(Tcw, P)
            u = (CAM[0] * c[:, 0] / c[:, 2] + CAM[2]).astype(np.float32)
            v = (CAM[1] * c[:, 1] / c[:, 2] + CAM[3]).astype(np.float32)
            ok &= ~(c[:, 2] < 0) & ~((u < bounds[0]) | (u > bounds[1])) & ~((v < bounds[2]) | (v > bounds[3]))
            if col is not None:
                out[:, col], out[:, col + 1] = u, v
                out[:, 4 + col // 2] = (np.float32(1.0) / c[:, 2]).astype(np.float32)
    return ok, out


@pytest.mark.parametrize("seed,stereo,edge,larger,direction,check", [(1, False, False, False, 0, True), (2, True, False, False, 0, True),
                                                                    (3, False, True, True, 1, True), (4, True, True, False, 2, False),
                                                                    (6, True, False, True, 0, True)])
def test_lines_search_by_projection_last_frame(oracle, ref, seed, stereo, edge, larger, direction, check):
    c = tlp.make_case(seed, n_cur=150, n_last=130, stereo=stereo, theta_edge=edge)
    rng = np.random.default_rng(seed + 900)
    Tcw = _pose(rng, 6.0, rng.uniform(-0.2, 0.2, 3))
    # world end points whose projections are about the case's (depth = 1 / proj[:, 4:6])
    p = c["proj"].astype(np.float64)
    Rcw, tcw = Tcw[:, :3].astype(np.float64), Tcw[:, 3].astype(np.float64)
    ends = []
    for k, iz in ((0, 4), (2, 5)):
        z = 1.0 / p[:, iz]
        cam_pt = np.stack([(p[:, k] - CAM[2]) / CAM[0] * z, (p[:, k + 1] - CAM[3]) / CAM[1] * z, z], 1)
        ends.append((cam_pt - tcw) @ Rcw)
    xyz_w = np.concatenate(ends, 1).astype(np.float32)
    bounds = _f32([0.0, tlp.W, 0.0, tlp.H])
    ok, proj = _project_lines(Tcw, xyz_w, bounds)
    c2 = dict(c, proj=proj, valid=(c["valid"].astype(bool) & ok).astype(np.uint8))
    want_n, want = tlp.oracle_ff(oracle, c2, larger, direction, 0.8, check)
    mb = 0.08
    twc = -(Rcw.T @ tcw)
    dz = (0.0, 1.0, -1.0)[direction]
    Tlw = np.concatenate([np.eye(3), (-twc + np.array([0, 0, dz]))[:, None]], axis=1).astype(np.float32)
    F, keep = _line_view(c)
    got = np.full(len(c["kl"]), -7, np.int32)
    fn = ref.ref_lines_search_by_projection_ff
    fn.restype = _i
    fn.argtypes = [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]
    got_n = fn(ctypes.byref(F), tlp._p(c["occupied"]), _p(bounds), mb, _p(Tcw), _p(Tlw), _p(CAM), len(c["valid"]),
               tlp._p(c["valid"]), _p(xyz_w), tlp._p(c["octave"]), tlp._p(c["angle"]), tlp._p(c["ldesc"]), tlp._p(c["has_obs"]),
               int(larger), 0, 0.8, int(check), _p(got))
    assert got_n == want_n > 15
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- Frame::ComputeStereoMatches
from tests import test_stereo as tst  # noqa: E402


@pytest.mark.parametrize("name,nfeatures", [("urban1", 2000), ("shift17", 1000), ("swapped", 1000)])
def test_frame_compute_stereo_matches(oracle, ref, name, nfeatures):
    """The reference's own extractor and Frame::ComputeStereoMatches on the pair against the oracle's extractor +
    oracle/stereo.c: the same mvuRight / mvDepth, bit for bit."""
    left, right = tst.pair(name)
    (kl, dl, pl), (kr, dr, pr) = tst.oracle_side(oracle, left, right, nfeatures)
    s, inv = tst.scale_tables()
    want_u, want_z, _, kept = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, s, inv, tst.MB, np.float32(tst.KITTI_BF))
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    cap = kl.shape[0] + 16
    u, z = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nr = ctypes.c_int()
    fn = ref.ref_frame_compute_stereo_matches
    fn.restype = _i
    fn.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp]
    n = fn(_p(left), _p(right), left.shape[1], left.shape[0], left.strides[0], nfeatures, tst.SCALE, tst.NLEVELS, 20, 7,
           float(tst.MB), float(np.float32(tst.KITTI_BF)), _p(u), _p(z), cap, ctypes.byref(nr))
    assert n == kl.shape[0] and nr.value == kr.shape[0]
    assert u[:n].tobytes() == want_u.tobytes()
    assert z[:n].tobytes() == want_z.tobytes()
    if name != "swapped":
        assert kept > 0.25 * n
