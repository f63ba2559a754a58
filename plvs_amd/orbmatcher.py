"""Host-side mirror of PLVS2::ORBmatcher for the search functions that run through
libplvs_hip.so (reference include/ORBmatcher.h, src/ORBmatcher.cc).

    matcher = ORBmatcher(nnratio=0.8, checkOri=True)
    nmatches, assigned = matcher.SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints)

`F` and `vpMapPoints` are FrameView / MapPointView: the arrays the reference function reads of
its Frame and MapPoint objects.  `assigned[i]` is the index (into vpMapPoints) of the map
point given to keypoint i, or -1 — what the reference writes into F.mvpMapPoints.
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class _FrameViewC(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("x", _vp), ("y", _vp), ("octave", _vp), ("u_right", _vp), ("desc", _vp),
                ("min_x", _f), ("min_y", _f), ("grid_w_inv", _f), ("grid_h_inv", _f), ("scale_factors", _vp)]


class _MapPointViewC(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int32), ("track_in_view", _vp), ("bad", _vp), ("proj_x", _vp), ("proj_y", _vp),
                ("proj_xr", _vp), ("view_cos", _vp), ("track_depth", _vp), ("level", _vp), ("desc", _vp),
                ("has_obs", _vp)]


@dataclass
class FrameView:
    x: np.ndarray               # mvKeysUn[i].pt.x
    y: np.ndarray
    octave: np.ndarray          # mvKeysUn[i].octave
    u_right: np.ndarray         # mvuRight[i]
    desc: np.ndarray            # mDescriptors [n,32]
    min_x: float                # mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
    min_y: float
    grid_w_inv: float
    grid_h_inv: float
    scale_factors: np.ndarray   # mvScaleFactors

    def as_c(self):
        self._keep = [np.ascontiguousarray(self.x, np.float32), np.ascontiguousarray(self.y, np.float32),
                      np.ascontiguousarray(self.octave, np.int32), np.ascontiguousarray(self.u_right, np.float32),
                      np.ascontiguousarray(self.desc, np.uint8), np.ascontiguousarray(self.scale_factors, np.float32)]
        k = self._keep
        return _FrameViewC(len(k[0]), _lib.np_ptr(k[0]), _lib.np_ptr(k[1]), _lib.np_ptr(k[2]), _lib.np_ptr(k[3]),
                           _lib.np_ptr(k[4]), self.min_x, self.min_y, self.grid_w_inv, self.grid_h_inv,
                           _lib.np_ptr(k[5]))


@dataclass
class MapPointView:
    track_in_view: np.ndarray   # mbTrackInView
    bad: np.ndarray             # isBad()
    proj_x: np.ndarray          # mTrackProjX / Y / XR
    proj_y: np.ndarray
    proj_xr: np.ndarray
    view_cos: np.ndarray        # mTrackViewCos
    track_depth: np.ndarray     # mTrackDepth
    level: np.ndarray           # mnTrackScaleLevel
    desc: np.ndarray            # GetDescriptor() [m,32]
    has_obs: np.ndarray = None  # Observations() > 0 (None: all)

    def as_c(self):
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self._keep = [u8(self.track_in_view), u8(self.bad), f32(self.proj_x), f32(self.proj_y), f32(self.proj_xr),
                      f32(self.view_cos), f32(self.track_depth), np.ascontiguousarray(self.level, np.int32),
                      u8(self.desc), None if self.has_obs is None else u8(self.has_obs)]
        k = self._keep
        return _MapPointViewC(len(k[0]), *[_lib.np_ptr(a) for a in k])


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 12

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(1, 32)
        d = np.zeros(1, np.int32)
        z = np.zeros(1, np.int32)
        _lib.check(_lib.lib.plvs_hip_hamming_pairs(_lib.np_ptr(a), 1, _lib.np_ptr(b), 1, _lib.np_ptr(z), _lib.np_ptr(z),
                                                   1, _lib.np_ptr(d)))
        return int(d[0])

    def SearchByProjection(self, F, vpMapPoints, th=1.0, bFarPoints=False, thFarPoints=50.0, occupied=None):
        fc, mc = F.as_c(), vpMapPoints.as_c()
        assigned = np.full(fc.n, -7, np.int32)
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        n = _i()
        f = _lib.lib.plvs_hip_orb_search_by_projection
        f.argtypes = [_vp, _vp, _f, _i, _f, _f, _vp, _vp, _vp]
        _lib.check(f(ctypes.byref(fc), ctypes.byref(mc), th, int(bFarPoints), thFarPoints, self.mfNNratio,
                     _lib.np_ptr(occ), _lib.np_ptr(assigned), ctypes.byref(n)))
        return n.value, assigned


class _LastFrameViewC(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("valid", _vp), ("u", _vp), ("v", _vp), ("invz", _vp), ("octave", _vp),
                ("angle", _vp), ("desc", _vp), ("has_obs", _vp)]


@dataclass
class LastFrameView:
    valid: np.ndarray     # LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i]
    u: np.ndarray         # projection of the map point into the current frame
    v: np.ndarray
    invz: np.ndarray      # 1 / depth in the current camera
    octave: np.ndarray    # LastFrame.mvKeys[i].octave
    angle: np.ndarray     # LastFrame.mvKeysUn[i].angle
    desc: np.ndarray      # pMP->GetDescriptor() [n,32]
    has_obs: np.ndarray = None

    def as_c(self):
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self._keep = [u8(self.valid), f32(self.u), f32(self.v), f32(self.invz),
                      np.ascontiguousarray(self.octave, np.int32), f32(self.angle), u8(self.desc),
                      None if self.has_obs is None else u8(self.has_obs)]
        k = self._keep
        return _LastFrameViewC(len(k[0]), *[_lib.np_ptr(a) for a in k])


def _search_ff(self, CurrentFrame, cur_angle, max_x, max_y, mbf, LastFrame, th, bForward=False, bBackward=False,
               occupied=None):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1774)."""
    fc, lc = CurrentFrame.as_c(), LastFrame.as_c()
    ang = np.ascontiguousarray(cur_angle, np.float32)
    assigned = np.full(fc.n, -7, np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n = _i()
    f = _lib.lib.plvs_hip_orb_search_by_projection_ff
    f.argtypes = [_vp, _vp, _f, _f, _f, _vp, _f, _i, _i, _i, _vp, _vp, _vp]
    _lib.check(f(ctypes.byref(fc), _lib.np_ptr(ang), max_x, max_y, mbf, ctypes.byref(lc), th, int(bForward),
                 int(bBackward), int(self.mbCheckOrientation), _lib.np_ptr(occ), _lib.np_ptr(assigned), ctypes.byref(n)))
    return n.value, assigned


ORBmatcher.SearchByProjectionLastFrame = _search_ff


class _FeatVecC(ctypes.Structure):
    _fields_ = [("nnodes", ctypes.c_int32), ("node_id", _vp), ("offset", _vp), ("index", _vp)]


class FeatureVector:
    """DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) from a {node id: [feature
    indices]} mapping; traversed in ascending node id like the std::map."""

    def __init__(self, nodes):
        ids = sorted(nodes)
        self.node_id = np.array(ids, np.uint32)
        lens = [len(nodes[k]) for k in ids]
        self.offset = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        self.index = (np.concatenate([np.asarray(nodes[k], np.uint32) for k in ids]) if ids
                      else np.zeros(0, np.uint32)).astype(np.uint32)

    def as_c(self):
        return _FeatVecC(len(self.node_id), _lib.np_ptr(self.node_id), _lib.np_ptr(self.offset), _lib.np_ptr(self.index))


def _search_bow(self, kf_featvec, kf_desc, kf_valid, kf_angle, f_featvec, f_desc, f_angle):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (src/ORBmatcher.cc:300).  kf_valid[i]: the key frame's
    keypoint i holds a good map point.  -> (nmatches, assigned [F.N]: key-frame keypoint index or -1)."""
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    fd = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    ka = np.ascontiguousarray(kf_angle, np.float32)
    fa = np.ascontiguousarray(f_angle, np.float32)
    assigned = np.full(fd.shape[0], -7, np.int32)
    kc, fc = kf_featvec.as_c(), f_featvec.as_c()
    n = _i()
    f = _lib.lib.plvs_hip_orb_search_by_bow
    f.argtypes = [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _vp, _vp]
    _lib.check(f(ctypes.byref(kc), _lib.np_ptr(kd), kd.shape[0], _lib.np_ptr(kv), _lib.np_ptr(ka), ctypes.byref(fc),
                 _lib.np_ptr(fd), fd.shape[0], _lib.np_ptr(fa), self.mfNNratio, int(self.mbCheckOrientation),
                 _lib.np_ptr(assigned), ctypes.byref(n)))
    return n.value, assigned


ORBmatcher.SearchByBoW = _search_bow
