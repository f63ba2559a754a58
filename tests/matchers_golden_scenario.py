"""One fixed set of search-function cases — the cases of tests/test_orb_search.py, test_line_search.py and
test_line_proj_search.py — run on the reference's own ORBmatcher.cc / LineMatcher.cc compiled unmodified
(scripts/make_matchers_golden.py -> tests/golden/matchers_reference_digests.json), on the oracle and on the HIP path
(tests/test_matchers_golden_reference.py): the device is checked against what the reference's code itself returned."""
import ctypes
import hashlib

import numpy as np

from tests import test_line_proj_search as tlp
from tests import test_line_search as tls
from tests import test_orb_search as tos

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


CASES = ([("M3 SearchByProjection(F, MapPoints)", s, th, far) for s, th, far in ((1, 1.0, False), (2, 3.0, False), (5, 1.0, True), (8, 5.0, True))] +
         [("M4 SearchByBoW", s, r, c) for s, r, c in ((1, 0.7, 1), (2, 0.7, 0), (3, 0.9, 1), (4, 0.5, 1))] +
         [("M7 SearchByKnn(F, LastF)", s, r, c) for s, r, c in ((1, 0.8, True), (2, 0.7, True), (3, 0.9, False), (4, 0.8, True))] +
         [("M7 SearchByKnn(pKF, F)", s, r, c) for s, r, c in ((1, 0.8, True), (2, 0.7, True), (3, 0.9, False))] +
         [("M7 SearchStereoMatchesByKnn", s, r, c, dd) for s, r, c, dd in ((1, 0.8, True, 50), (2, 0.7, True, 60), (3, 0.9, False, 50), (4, 0.8, True, 256))] +
         [("M9 SearchByProjection(F, MapLines)", s, st, e, lg, r) for s, st, e, lg, r in ((1, False, False, False, 0.8), (2, True, False, False, 0.8),
                                                                                      (3, False, True, True, 0.9), (4, True, True, False, 0.7))] +
         [("M5 Frame::ComputeStereoMatches", name, nf) for name, nf in (("urban1", 2000), ("shift17", 1000))] +
         [("G Frame::UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid / UndistortKeyLines", ci) for ci in range(5)])


def _digest(n, *arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return dict(matches=int(n), sha1=h.hexdigest())


def _knn_ref(ref, name):
    fn = getattr(ref, name)
    fn.restype = _i
    fn.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _f, _i, _vp]
    return fn


def run_case(case, backend, oracle=None, ref=None):
    """backend: 'ref' (ctypes handle of oracle/_ref/libmatchers_ref.so), 'oracle', 'hip'."""
    kind = case[0]
    if kind.startswith("M3"):
        _, seed, th, far = case
        F, M, occ = tos.make_case(seed)
        if backend == "oracle":
            n, a = tos.oracle_search(oracle.lib, F, M, th, far, 40.0, 0.8, occ)
        elif backend == "hip":
            from plvs_amd.orbmatcher import ORBmatcher
            n, a = ORBmatcher(0.8, True).SearchByProjection(F, M, th, far, 40.0, occupied=occ)
        else:
            fc, mc = F.as_c(), M.as_c()
            a = np.full(fc.n, -7, np.int32)
            fn = ref.ref_orb_search_by_projection
            fn.argtypes = [_vp, _vp, _f, _i, _f, _f, _vp, _vp]
            fn.restype = _i
            n = fn(ctypes.byref(fc), ctypes.byref(mc), th, int(far), 40.0, 0.8, _p(occ), _p(a))
        return _digest(n, a)
    if kind.startswith("M4"):
        _, seed, ratio, check = case
        KV, kd, kv, ka, FV, fd, fa = tos.make_bow_case(seed)
        if backend == "oracle":
            n, a = tos.oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, ratio, check)
        elif backend == "hip":
            from plvs_amd.orbmatcher import ORBmatcher
            n, a = ORBmatcher(ratio, bool(check)).SearchByBoW(KV, kd, kv, ka, FV, fd, fa)
        else:
            kc, fc = KV.as_c(), FV.as_c()
            a = np.full(fd.shape[0], -7, np.int32)
            fn = ref.ref_orb_search_by_bow
            fn.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _vp]
            fn.restype = _i
            n = fn(ctypes.byref(kc), _p(kd), kd.shape[0], _p(kv), _p(ka), ctypes.byref(fc), _p(fd), fd.shape[0], _p(fa), ratio, check, _p(a))
        return _digest(n, a)
    if kind == "M7 SearchByKnn(F, LastF)" or kind == "M7 SearchByKnn(pKF, F)":
        _, seed, ratio, check = case
        kf = kind.endswith("(pKF, F)")
        c = (tls.make_case(10 + seed, n_last=120 + 7 * seed, n_cur=100 + 3 * seed, rot=0.4 * seed) if kf else
             tls.make_case(seed, n_last=90 + 7 * seed, n_cur=100 + 3 * seed, rot=0.3 * seed))
        if backend == "oracle":
            n, a = tls.run(tls.oracle_kf_fn(oracle) if kf else tls.oracle_fn(oracle), c, ratio, check)
        elif backend == "hip":
            from plvs_amd.linematcher import LineMatcher
            m = LineMatcher(ratio, check)
            n, a = (m.SearchByKnn if kf else m.SearchByKnnLastFrame)(*c)
        else:
            n, a = tls.run(_knn_ref(ref, "ref_lines_search_by_knn_kf" if kf else "ref_lines_search_by_knn"), c, ratio, check)
        return _digest(n, a)
    if kind.startswith("M7 SearchStereo"):
        _, seed, ratio, check, dd = case
        c = tls.make_stereo_case(seed, n_left=100 + 9 * seed, n_right=90 + 5 * seed)
        if backend == "oracle":
            n, q, t, d, v = tls.oracle_stereo(oracle, c, ratio, check, dd)
        elif backend == "hip":
            from plvs_amd.linematcher import LineMatcher
            n, m, v = LineMatcher(ratio, check).SearchStereoMatchesByKnn(*c, descriptorDist=dd)
            q, t, d = m["queryIdx"], m["trainIdx"], m["distance"]
        else:
            left, ang_l, oct_l, right, ang_r, oct_r = c
            cap = max(right.shape[0], 1)
            q, t = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            d, v = np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
            n_out = ctypes.c_int()
            fn = ref.ref_lines_search_stereo_by_knn
            fn.restype = _i
            fn.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]
            n = fn(_p(left), left.shape[0], _p(ang_l), _p(oct_l), _p(right), right.shape[0], _p(ang_r), _p(oct_r), ratio, int(check), dd,
                   _p(q), _p(t), _p(d), _p(v), ctypes.byref(n_out))
            k = n_out.value
            q, t, d, v = q[:k], t[:k], d[:k], v[:k]
        return _digest(n, np.asarray(q, np.int32), np.asarray(t, np.int32), np.asarray(d, np.float32), np.asarray(v, np.uint8))
    if kind.startswith("M9"):
        _, seed, stereo, edge, larger, ratio = case
        c = tlp.make_case(seed, n_cur=150, n_last=130, stereo=stereo, theta_edge=edge)
        if backend == "oracle":
            n, a = tlp.oracle_map(oracle, c, larger, ratio)
        else:
            from plvs_amd.linematcher import LineMatcher, line_frame_view
            if backend == "hip":
                view = line_frame_view(c["kl"], c["desc"], tlp.SCALE, tlp.INV_SIGMA2, tlp.MAX_DIAG, c["urs"], c["ure"], c["bf"])
                n, a = LineMatcher(ratio, True).SearchByProjection(view, c["valid"], c["proj_map"], c["octave"], c["ldesc"],
                                                                   occupied=c["occupied"], has_obs=c["has_obs"], bLargerSearch=larger)
            else:
                F, keep = line_frame_view(c["kl"], c["desc"], tlp.SCALE, tlp.INV_SIGMA2, tlp.MAX_DIAG, u_right_start=c["urs"],
                                          u_right_end=c["ure"], bf=c["bf"])
                a = np.full(len(c["kl"]), -7, np.int32)
                fn = ref.ref_lines_search_by_projection
                fn.restype = _i
                fn.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]
                n = fn(ctypes.byref(F), tlp._p(c["occupied"]), len(c["valid"]), tlp._p(c["valid"]), tlp._p(c["proj_map"]),
                       tlp._p(c["octave"]), tlp._p(c["ldesc"]), tlp._p(c["has_obs"]), int(larger), ratio, _p(a))
        return _digest(n, np.asarray(a, np.int32))
    if kind.startswith("M5"):   # the reference runs its OWN extractor on the pair, then Frame::ComputeStereoMatches
        from tests import test_stereo as tst
        _, name, nf = case
        left, right = tst.pair(name)
        left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        if backend == "ref":
            cap = 2 * nf + 64
            u, z = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
            nr = ctypes.c_int()
            fn = ref.ref_frame_compute_stereo_matches
            fn.restype = _i
            fn.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp]
            n = fn(_p(left), _p(right), left.shape[1], left.shape[0], left.strides[0], nf, tst.SCALE, tst.NLEVELS, 20, 7,
                   float(tst.MB), float(np.float32(tst.KITTI_BF)), _p(u), _p(z), cap, ctypes.byref(nr))
            u, z = u[:n], z[:n]
        elif backend == "oracle":
            (kl, dl, pl), (kr, dr, pr) = tst.oracle_side(oracle, left, right, nf)
            sc, inv = tst.scale_tables()
            u, z, _, _ = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, sc, inv, tst.MB, np.float32(tst.KITTI_BF))
        else:
            from plvs_amd.orb import ORBextractor
            from plvs_amd.stereo import StereoMatcher
            exl, exr = ORBextractor(nf, tst.SCALE, tst.NLEVELS, 20, 7), ORBextractor(nf, tst.SCALE, tst.NLEVELS, 20, 7)
            _, hkl, hdl = exl(left)
            _, hkr, hdr = exr(right)
            u, z = StereoMatcher(exl, exr).ComputeStereoMatches(hkl, hdl, hkr, hdr, tst.MB, np.float32(tst.KITTI_BF))
        return _digest(int((np.asarray(u) >= 0).sum()), np.asarray(u, np.float32), np.asarray(z, np.float32))
    if kind.startswith("G "):   # the glue of Frame::Frame between extraction and search, per calibration of tests/test_frame_glue.py
        from tests import test_frame_glue as tfg
        _, ci = case
        side = (tfg._Side(ref, "ref_frame_") if backend == "ref" else
                tfg._Side(oracle.lib, "oracle_frame_") if backend == "oracle" else tfg._Hip())
        K, D = tfg.CALIBS[ci]
        k = tfg._keypoints(70 + ci, 2100 if ci == 0 else 300)
        un = side.undistort_keypoints(k, K, D)
        b = np.asarray(side.bounds(640, 480, K, D), np.float32)
        iw, ih = np.float32(64.0) / np.float32(b[1] - b[0]), np.float32(48.0) / np.float32(b[3] - b[2])
        start, items = side.grid(un, float(b[0]), float(b[2]), float(iw), float(ih))
        kl = tfg._keylines(120 + ci, 160)
        lu, kept = side.undistort_keylines(kl, K, D, b)
        return _digest(len(items), un, b, np.asarray(start, np.int32), np.asarray(items, np.int32), np.ascontiguousarray(lu),
                       np.asarray(kept, np.int32))
    raise ValueError(kind)


def run(backend, oracle=None, ref=None):
    return [dict(case=list(c), **run_case(c, backend, oracle=oracle, ref=ref)) for c in CASES]
