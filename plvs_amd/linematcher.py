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


class LineMatcher:
    TH_HIGH, TH_LOW, TH_LOW_STEREO, HISTO_LENGTH = 110, 60, 50, 12     # src/LineMatcher.cc:87-90

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
