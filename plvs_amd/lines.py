"""Host-side mirror of PLVS2::LineExtractor (include/LineExtractor.h:48-83).

    LineExtractor(numLinefeatures, opts)        opts: numOctaves, scale, min_length, lineFitErrThreshold
    extractor(image, keylines, descriptors)     -> here: keylines, descriptors = extractor(image)

EDLines + LBD, i.e. the reference's default (Line.LSD.on: 0).  All compute and
the sequential host stages live in libplvs_hip.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib

KEYLINE_DTYPE = np.dtype([("angle", np.float32), ("class_id", np.int32), ("octave", np.int32), ("pt_x", np.float32),
                          ("pt_y", np.float32), ("response", np.float32), ("size", np.float32),
                          ("startPointX", np.float32), ("startPointY", np.float32), ("endPointX", np.float32),
                          ("endPointY", np.float32), ("sPointInOctaveX", np.float32), ("sPointInOctaveY", np.float32),
                          ("ePointInOctaveX", np.float32), ("ePointInOctaveY", np.float32),
                          ("lineLength", np.float32), ("numOfPixels", np.int32)])
assert KEYLINE_DTYPE.itemsize == 68

_vp, _i, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double
L = _lib.lib
L.plvs_hip_lines_create.argtypes = [_i, _i, _f, _d, _d, ctypes.POINTER(_vp)]
L.plvs_hip_lines_destroy.argtypes = [_vp]
L.plvs_hip_lines_extract.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]
L.plvs_hip_lines_extract_dev.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]
L.plvs_hip_lines_last_stage_ms.argtypes = [_vp, _vp, _i]
L.plvs_hip_lines_octave_size.argtypes = [_vp, _i, _vp, _vp]
L.plvs_hip_lines_download_map.argtypes = [_vp, _i, _i, _vp]
L.plvs_hip_lines_num_in_octave.argtypes = [_vp, _i]


class LSDOptions:
    """The subset of cv::line_descriptor_c::LSDDetectorC::LSDOptions the EDLines path reads."""

    def __init__(self, numOctaves=3, scale=1.2, min_length=0.02, lineFitErrThreshold=1.6):
        self.numOctaves, self.scale = numOctaves, scale
        self.min_length, self.lineFitErrThreshold = min_length, lineFitErrThreshold


class LineExtractor:
    skUseLsdExtractor = False

    def __init__(self, numLinefeatures, opts=None):
        opts = opts or LSDOptions()
        if self.skUseLsdExtractor:
            raise NotImplementedError("the LSD detector (Line.LSD.on: 1) is not on the accelerated path")
        self.opts = opts
        self._h = _vp()
        _lib.check(L.plvs_hip_lines_create(numLinefeatures, opts.numOctaves, opts.scale, opts.min_length,
                                           opts.lineFitErrThreshold, ctypes.byref(self._h)))
        self._cap = 4096
        self._kl = np.zeros(self._cap, KEYLINE_DTYPE)
        self._desc = np.zeros((self._cap, 32), np.uint8)

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_lines_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, image):
        """-> (keylines [structured array of KeyLine fields], descriptors [n,32] u8)"""
        n = _i()
        if isinstance(image, torch.Tensor):
            assert image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2
            torch.cuda.current_stream().synchronize()
            h, w = image.shape
            rc = L.plvs_hip_lines_extract_dev(self._h, _vp(image.data_ptr()), w, h, image.stride(0),
                                              _lib.np_ptr(self._kl), _lib.np_ptr(self._desc), self._cap, ctypes.byref(n))
        else:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            h, w = image.shape
            rc = L.plvs_hip_lines_extract(self._h, _lib.np_ptr(image), w, h, w, _lib.np_ptr(self._kl),
                                          _lib.np_ptr(self._desc), self._cap, ctypes.byref(n))
        _lib.check(rc)
        if n.value == 0:
            print("LineExtractor::detectLineFeatures() - no lines! **********")
        return self._kl[:n.value].copy(), self._desc[:n.value].copy()

    def SetGaussianPyramid(self, orb_extractor):
        """LineExtractor::SetGaussianPyramid as Frame::PrecomputeGaussianPyramid uses it (src/Frame.cc:848):
        the octaves become the levels of `orb_extractor`'s device pyramid; None restores the own chain."""
        self._shared = orb_extractor                       # keep the handle alive
        L.plvs_hip_lines_set_gaussian_pyramid.argtypes = [_vp, _vp]
        _lib.check(L.plvs_hip_lines_set_gaussian_pyramid(self._h, orb_extractor._h if orb_extractor else None))

    def stage_ms(self):
        ms = (ctypes.c_double * 6)()
        _lib.check(L.plvs_hip_lines_last_stage_ms(self._h, ms, 6))
        return dict(zip(["device_maps", "host_link_fit_group", "lbd", "host_edge_drawing",
                         "host_line_fit", "host_group_select"], list(ms)))

    def octave_map(self, octave, which):
        """which: 'blur' (u8), 'dx', 'dy' (s16), 'gd' (u16 packed)"""
        w, h = _i(), _i()
        _lib.check(L.plvs_hip_lines_octave_size(self._h, octave, ctypes.byref(w), ctypes.byref(h)))
        code = {"blur": 0, "dx": 1, "dy": 2, "gd": 3}[which]
        out = np.zeros((h.value, w.value), {0: np.uint8, 1: np.int16, 2: np.int16, 3: np.uint16}[code])
        _lib.check(L.plvs_hip_lines_download_map(self._h, octave, code, _lib.np_ptr(out)))
        return out

    def num_in_octave(self, octave):
        return L.plvs_hip_lines_num_in_octave(self._h, octave)
