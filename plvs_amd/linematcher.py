"""Host-side mirror of PLVS2::LineMatcher for the descriptor searches that run through libplvs_hip.so
(reference include/LineMatcher.h, src/LineMatcher.cc).  Lines are handed over as the arrays the reference
functions read: LBD descriptors [n,32] u8, mvKeyLinesUn[i].angle (radians), .octave, and the validity of
the map line attached to each query line."""
import ctypes

import numpy as np

from . import _lib

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class LineFrameView(ctypes.Structure):
    """plvs_line_frame_view (include/plvs_hip.h): what the guided line searches read of a Frame."""
    _fields_ = [("n", ctypes.c_int32), ("keylines_un", _vp), ("descriptors", _vp), ("u_right_start", _vp),
                ("u_right_end", _vp), ("bf", _f), ("n_levels", ctypes.c_int32), ("line_scale_factors", _vp),
                ("line_inv_level_sigma2", _vp), ("max_diag", _f)]


def line_frame_view(keylines_un, descriptors, scale_factors, inv_level_sigma2, max_diag, u_right_start=None,
                    u_right_end=None, bf=0.0):
    """-> (view, keep-alive tuple).  keylines_un: structured array with the KeyLine layout (68 bytes)."""
    kl = np.ascontiguousarray(keylines_un)
    assert kl.dtype.itemsize == 68
    d = _u8(descriptors).reshape(-1, 32)
    sf, s2 = _f32(scale_factors), _f32(inv_level_sigma2)
    us = None if u_right_start is None else _f32(u_right_start)
    ue = None if u_right_end is None else _f32(u_right_end)
    v = LineFrameView(kl.shape[0], _lib.np_ptr(kl), _lib.np_ptr(d), _lib.np_ptr(us), _lib.np_ptr(ue), float(bf),
                      sf.shape[0], _lib.np_ptr(sf), _lib.np_ptr(s2), float(max_diag))
    return v, (kl, d, sf, s2, us, ue)


class LineMatcher:
    TH_HIGH, TH_LOW, TH_LOW_STEREO, HISTO_LENGTH = 110, 60, 50, 12     # src/LineMatcher.cc:87-90

    def SearchByProjectionLastFrame(self, view, valid, proj, octave, angle, desc, occupied=None, has_obs=None,
                                    bLargerSearch=False, direction=0):
        """SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono),
        src/LineMatcher.cc:837.  proj [n_last, 6] = uS, vS, uE, vE, invSz, invEz of the last frame's map lines
        in the current frame.  -> (nmatches, assigned [Nlines]: last-frame line index or -1)."""
        F, keep = view
        va, pr = _u8(valid), _f32(proj).reshape(-1, 6)
        oc, an, de = np.ascontiguousarray(octave, np.int32), _f32(angle), _u8(desc).reshape(-1, 32)
        occ = None if occupied is None else _u8(occupied)
        ho = None if has_obs is None else _u8(has_obs)
        assigned = np.full(max(F.n, 1), -7, np.int32)
        n = _i()
        f = _lib.lib.plvs_hip_lines_search_by_projection_ff
        f.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp]
        _lib.check(f(ctypes.byref(F), _lib.np_ptr(occ), va.shape[0], _lib.np_ptr(va), _lib.np_ptr(pr), _lib.np_ptr(oc),
                     _lib.np_ptr(an), _lib.np_ptr(de), _lib.np_ptr(ho), int(bLargerSearch), int(direction),
                     self.mfNNratio, int(self.mbCheckOrientation), _lib.np_ptr(assigned), ctypes.byref(n)))
        return n.value, assigned[:F.n]

    def SearchByProjection(self, view, in_view, proj, level, desc, occupied=None, has_obs=None, bLargerSearch=False):
        """SearchByProjection(Frame& F, const std::vector<MapLinePtr>&, bLargerSearch), src/LineMatcher.cc:1286.
        proj [n_map, 6] = mTrackProjStartX, StartY, EndX, EndY, mTrackStartDepth, mTrackEndDepth (the DEPTHS: the stereo
        gate divides mbf by them, :1419-1423).  -> (nmatches, assigned [Nlines]: map-line index or -1)."""
        F, keep = view
        iv, pr = _u8(in_view), _f32(proj).reshape(-1, 6)
        lv, de = np.ascontiguousarray(level, np.int32), _u8(desc).reshape(-1, 32)
        occ = None if occupied is None else _u8(occupied)
        ho = None if has_obs is None else _u8(has_obs)
        assigned = np.full(max(F.n, 1), -7, np.int32)
        n = _i()
        f = _lib.lib.plvs_hip_lines_search_by_projection
        f.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp]
        _lib.check(f(ctypes.byref(F), _lib.np_ptr(occ), iv.shape[0], _lib.np_ptr(iv), _lib.np_ptr(pr), _lib.np_ptr(lv),
                     _lib.np_ptr(de), _lib.np_ptr(ho), int(bLargerSearch), self.mfNNratio, _lib.np_ptr(assigned),
                     ctypes.byref(n)))
        return n.value, assigned[:F.n]

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    def _frame_side(self, fn, desc_q, valid_q, angle_q, desc_t, angle_t):
        dq, dt = _u8(desc_q).reshape(-1, 32), _u8(desc_t).reshape(-1, 32)
        vq, aq, at = _u8(valid_q), _f32(angle_q), _f32(angle_t)
        assigned = np.full(dt.shape[0], -7, np.int32)
        n = _i()
        fn.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _f, _i, _vp, _vp]
        _lib.check(fn(_lib.np_ptr(dq), dq.shape[0], _lib.np_ptr(vq), _lib.np_ptr(aq), _lib.np_ptr(dt), dt.shape[0],
                      _lib.np_ptr(at), self.mfNNratio, int(self.mbCheckOrientation), _lib.np_ptr(assigned),
                      ctypes.byref(n)))
        return n.value, assigned

    def SearchByKnnLastFrame(self, desc_last, valid_last, angle_last, desc_cur, angle_cur):
        """SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame), src/LineMatcher.cc:303."""
        return self._frame_side(_lib.lib.plvs_hip_lines_search_by_knn, desc_last, valid_last, angle_last, desc_cur,
                                angle_cur)

    def SearchByKnn(self, desc_kf, valid_kf, angle_kf, desc_f, angle_f):
        """SearchByKnn(KeyFramePtr& pKF, const Frame& F, vpMapLineMatches), src/LineMatcher.cc:156.
        -> (nmatches, assigned [F.Nlines]: key-frame line index or -1)."""
        return self._frame_side(_lib.lib.plvs_hip_lines_search_by_knn_kf, desc_kf, valid_kf, angle_kf, desc_f, angle_f)

    def SearchStereoMatchesByKnn(self, desc_left, angle_left, octave_left, desc_right, angle_right, octave_right,
                                 descriptorDist=None):
        """src/LineMatcher.cc:454.  -> (numValidMatches, vMatches [k] of (queryIdx, trainIdx, distance),
        vValidMatches [k] bool)."""
        if descriptorDist is None:
            descriptorDist = self.TH_LOW_STEREO
        dl, dr = _u8(desc_left).reshape(-1, 32), _u8(desc_right).reshape(-1, 32)
        al, ar = _f32(angle_left), _f32(angle_right)
        ol, orr = np.ascontiguousarray(octave_left, np.int32), np.ascontiguousarray(octave_right, np.int32)
        cap = max(dr.shape[0], 1)
        mq, mt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        md, mv = np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
        n_out, n = _i(), _i()
        f = _lib.lib.plvs_hip_lines_search_stereo_by_knn
        f.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]
        _lib.check(f(_lib.np_ptr(dl), dl.shape[0], _lib.np_ptr(al), _lib.np_ptr(ol), _lib.np_ptr(dr), dr.shape[0],
                     _lib.np_ptr(ar), _lib.np_ptr(orr), self.mfNNratio, int(self.mbCheckOrientation),
                     int(descriptorDist), _lib.np_ptr(mq), _lib.np_ptr(mt), _lib.np_ptr(md), _lib.np_ptr(mv), cap,
                     ctypes.byref(n_out), ctypes.byref(n)))
        k = n_out.value
        matches = np.zeros(k, np.dtype([("queryIdx", np.int32), ("trainIdx", np.int32), ("distance", np.float32)]))
        matches["queryIdx"], matches["trainIdx"], matches["distance"] = mq[:k], mt[:k], md[:k]
        return n.value, matches, mv[:k].astype(bool)
