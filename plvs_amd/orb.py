"""Host-side mirror of PLVS2::ORBextractor (include/ORBextractor.h:68-170).

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
    monoIndex = extractor(image, mask, keypoints, descriptors, vLappingArea)
    GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors /
    GetScaleSigmaSquares / GetInverseScaleSigmaSquares

Same names, argument meaning and error behaviour as the reference; all compute
runs in libplvs_hip.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib

KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])
assert KP_DTYPE.itemsize == 28

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L = _lib.lib
L.plvs_hip_orb_create.argtypes = [_i, _f, _i, _i, _i, ctypes.POINTER(_vp)]
L.plvs_hip_orb_destroy.argtypes = [_vp]
L.plvs_hip_orb_get_levels.argtypes = [_vp]
L.plvs_hip_orb_get_scale_factor.argtypes = [_vp]
L.plvs_hip_orb_get_scale_factor.restype = _f
L.plvs_hip_orb_get_scale_tables.argtypes = [_vp, _vp, _vp, _vp, _vp]
L.plvs_hip_orb_features_per_level.argtypes = [_vp, _vp]
L.plvs_hip_orb_extract.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]
L.plvs_hip_orb_extract_dev.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]
L.plvs_hip_orb_last_stage_ms.argtypes = [_vp, _vp, _i]
L.plvs_hip_orb_level_size.argtypes = [_vp, _i, _vp, _vp]
L.plvs_hip_orb_download_level.argtypes = [_vp, _i, _i, _vp]
L.plvs_hip_orb_last_candidates.argtypes = [_vp, _i, _vp, _i, _vp]


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST):
        self._h = _vp()
        _lib.check(L.plvs_hip_orb_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, ctypes.byref(self._h)))
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._cap = nfeatures * 2 + 1024
        self._kps = np.zeros(self._cap, KP_DTYPE)
        self._desc = np.zeros((self._cap, 32), np.uint8)

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """-> (monoIndex, keypoints [structured array of cv::KeyPoint fields], descriptors [n,32] u8).
        Returns (-1, empty, empty) on an empty image, like the reference."""
        n, mono = _i(), _i()
        if isinstance(image, torch.Tensor):
            assert image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2
            torch.cuda.current_stream().synchronize()
            h, w = image.shape
            rc = L.plvs_hip_orb_extract_dev(self._h, _vp(image.data_ptr()), w, h, image.stride(0), vLappingArea[0],
                                            vLappingArea[1], _lib.np_ptr(self._kps), _lib.np_ptr(self._desc),
                                            self._cap, ctypes.byref(n), ctypes.byref(mono))
        else:
            image = np.asarray(image)
            if image.size == 0:
                return -1, self._kps[:0].copy(), self._desc[:0].copy()
            assert image.dtype == np.uint8 and image.ndim == 2, "image.type() == CV_8UC1"
            image = np.ascontiguousarray(image)
            h, w = image.shape
            rc = L.plvs_hip_orb_extract(self._h, _lib.np_ptr(image), w, h, w, vLappingArea[0], vLappingArea[1],
                                        _lib.np_ptr(self._kps), _lib.np_ptr(self._desc), self._cap,
                                        ctypes.byref(n), ctypes.byref(mono))
        if rc == _lib.PLVS_ERR_EMPTY:
            return -1, self._kps[:0].copy(), self._desc[:0].copy()
        _lib.check(rc)
        return mono.value, self._kps[:n.value].copy(), self._desc[:n.value].copy()

    # -- getters (ORBextractor.h:90-113)
    def GetLevels(self):
        return L.plvs_hip_orb_get_levels(self._h)

    def GetScaleFactor(self):
        return L.plvs_hip_orb_get_scale_factor(self._h)

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        _lib.check(L.plvs_hip_orb_get_scale_tables(self._h, *[_lib.np_ptr(x) for x in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        _lib.check(L.plvs_hip_orb_features_per_level(self._h, _lib.np_ptr(out)))
        return out

    # -- parity / debug accessors
    def stage_ms(self):
        ms = (ctypes.c_double * 8)()
        _lib.check(L.plvs_hip_orb_last_stage_ms(self._h, ms, 8))
        return dict(zip(["pyramid_fast_cells", "host_quadtree", "orientation", "cossin_descriptors", "packing"], list(ms)[:5]))

    def level(self, level, blurred=False):
        w, h = _i(), _i()
        _lib.check(L.plvs_hip_orb_level_size(self._h, level, ctypes.byref(w), ctypes.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _lib.check(L.plvs_hip_orb_download_level(self._h, level, int(blurred), _lib.np_ptr(out)))
        return out

    def candidates(self, level):
        n = _i()
        _lib.check(L.plvs_hip_orb_last_candidates(self._h, level, None, 0, ctypes.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        _lib.check(L.plvs_hip_orb_last_candidates(self._h, level, _lib.np_ptr(out), n.value, ctypes.byref(n)))
        return out[:n.value]
