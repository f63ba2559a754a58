"""Host-side mirror of Frame::ComputeStereoMatches (src/Frame.cc:1780-1975): sparse stereo matching of
the ORB keypoints of a rectified pair.  The arithmetic runs in libplvs_hip.so on the device pyramids the
two extractors hold; there is no CPU fallback."""
import ctypes

import numpy as np

from . import _lib
from .orb import KP_DTYPE

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L = _lib.lib
L.plvs_hip_stereo_create.argtypes = [_vp, _vp, ctypes.POINTER(_vp)]
L.plvs_hip_stereo_destroy.argtypes = [_vp]
L.plvs_hip_stereo_matches.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, ctypes.POINTER(_i)]


class StereoMatcher:
    """Bound to the left / right ORBextractor of a Frame (mpORBextractorLeft / mpORBextractorRight)."""

    def __init__(self, extractor_left, extractor_right):
        self._left, self._right = extractor_left, extractor_right       # keep the handles alive
        self._h = _vp()
        _lib.check(L.plvs_hip_stereo_create(extractor_left._h, extractor_right._h, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_stereo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ComputeStereoMatches(self, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mb, mbf):
        """-> (mvuRight, mvDepth) float32 [N], -1 where a keypoint has no stereo match."""
        kl = np.ascontiguousarray(mvKeys, dtype=KP_DTYPE)
        kr = np.ascontiguousarray(mvKeysRight, dtype=KP_DTYPE)
        dl = np.ascontiguousarray(mDescriptors, dtype=np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(mDescriptorsRight, dtype=np.uint8).reshape(-1, 32)
        if dl.shape[0] != kl.shape[0] or dr.shape[0] != kr.shape[0]:
            raise ValueError("one 32-byte descriptor per keypoint")
        u_right = np.full(kl.shape[0], -1.0, np.float32)
        depth = np.full(kl.shape[0], -1.0, np.float32)
        n = _i()
        _lib.check(L.plvs_hip_stereo_matches(self._h, _lib.np_ptr(kl), _lib.np_ptr(dl), kl.shape[0], _lib.np_ptr(kr),
                                             _lib.np_ptr(dr), kr.shape[0], mb, mbf, _lib.np_ptr(u_right),
                                             _lib.np_ptr(depth), ctypes.byref(n)))
        return u_right, depth
