"""Host-side mirror of PLVS2::LineExtractor (include/LineExtractor.h:48-83).

    LineExtractor(numLinefeatures, opts)        opts: numOctaves, scale, min_length, lineFitErrThreshold
    extractor(image, keylines, descriptors)     -> here: keylines, descriptors = extractor(image)

EDLines + LBD, i.e. the reference's default (Line.LSD.on: 0), or — LineExtractor.skUseLsdExtractor = True, Line.LSD.on: 1 —
the LSD detector in front of the same descriptor; its own interfaces are mirrored too:

    LSDDetectorC.createLSDDetectorC(opts).detect(image, scale, numOctaves, opts)   -> keylines
    createLineSegmentDetector(refine, scale, ...).detect(image)                    -> [n, 4] float32 (x1, y1, x2, y2)

All compute and the sequential host stages live in libplvs_hip.so.
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


L.plvs_hip_lsd_create.argtypes = [ctypes.POINTER(_vp)]
L.plvs_hip_lsd_destroy.argtypes = [_vp]
L.plvs_hip_lsd_default_options.argtypes = [_vp]
L.plvs_hip_lsd_segments.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]
L.plvs_hip_lsd_detect.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _vp, _d, _vp, _i, _vp]
L.plvs_hip_lsd_extract.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _d, _vp, _vp, _i, _vp]
L.plvs_hip_lsd_extract_dev.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _d, _vp, _vp, _i, _vp]
L.plvs_hip_lsd_last_stage_ms.argtypes = [_vp, _vp, _i]

LSD_REFINE_NONE, LSD_REFINE_STD, LSD_REFINE_ADV = 0, 1, 2


class _LsdOptionsC(ctypes.Structure):      # plvs_lsd_options
    _fields_ = [("refine", _i), ("scale", _d), ("sigma_scale", _d), ("quant", _d), ("ang_th", _d), ("log_eps", _d),
                ("density_th", _d), ("n_bins", _i)]


class LSDOptions:
    """cv::line_descriptor_c::LSDDetectorC::LSDOptions (descriptor_custom.hpp:928-957).  The EDLines path reads numOctaves,
    scale, min_length and lineFitErrThreshold; the LSD path all but the last.  Defaults: the struct's own — Tracking
    overrides scale with Line.scaleFactor and refine / log_eps / density_th with 1 / 1.0 / 0.6 unless the settings file
    says otherwise (src/Tracking.cc:1466-1485)."""

    def __init__(self, numOctaves=3, scale=1.2, min_length=0.02, lineFitErrThreshold=1.6, refine=LSD_REFINE_ADV,
                 sigma_scale=0.6, quant=2.0, ang_th=22.5, log_eps=0.0, density_th=0.7, n_bins=1024):
        self.numOctaves, self.scale = numOctaves, scale
        self.min_length, self.lineFitErrThreshold = min_length, lineFitErrThreshold
        self.refine, self.sigma_scale, self.quant, self.ang_th = refine, sigma_scale, quant, ang_th
        self.log_eps, self.density_th, self.n_bins = log_eps, density_th, n_bins

    def _c(self, scale=None):
        return _LsdOptionsC(int(self.refine), float(self.scale if scale is None else scale), float(self.sigma_scale),
                            float(self.quant), float(self.ang_th), float(self.log_eps), float(self.density_th), int(self.n_bins))


class _LsdHandle:
    def __init__(self):
        self._h = _vp()
        _lib.check(L.plvs_hip_lsd_create(ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_lsd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stage_ms(self):
        ms = (ctypes.c_double * 4)()
        _lib.check(L.plvs_hip_lsd_last_stage_ms(self._h, ms, 4))
        return dict(zip(["device_maps", "host_regions", "select_lbd"], list(ms)[:3]))


def _gray(image):
    image = np.ascontiguousarray(image, dtype=np.uint8)
    assert image.ndim == 2
    return image


class LineSegmentDetector(_LsdHandle):
    """cv::lsd::createLineSegmentDetector(...) / LineSegmentDetector::detect (lsd_custom.cpp:398-446)."""

    def __init__(self, refine=LSD_REFINE_STD, scale=0.8, sigma_scale=0.6, quant=2.0, ang_th=22.5, log_eps=0.0,
                 density_th=0.7, n_bins=1024):
        super().__init__()
        self._opt = _LsdOptionsC(refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins)

    def detect(self, image):
        image = _gray(image)
        h, w = image.shape
        cap, n = 4096, _i()
        while True:
            out = np.zeros((cap, 4), np.float32)
            _lib.check(L.plvs_hip_lsd_segments(self._h, _lib.np_ptr(image), w, h, w, ctypes.byref(self._opt),
                                               _lib.np_ptr(out), cap, ctypes.byref(n)))
            if n.value <= cap:
                return out[:n.value]
            cap = n.value


createLineSegmentDetector = LineSegmentDetector


class LSDDetectorC(_LsdHandle):
    """cv::line_descriptor_c::LSDDetectorC (LSDDetector_custom.cpp:50-298)."""

    def __init__(self, opts=None):
        super().__init__()
        self.opts = opts or LSDOptions(numOctaves=1, scale=0.8)

    @classmethod
    def createLSDDetectorC(cls, opts=None):
        return cls(opts)

    def detect(self, image, scale, numOctaves, opts=None):
        """detect(image, keylines, scale, numOctaves, opts): `scale` the pyramid's, opts.scale the detector's own."""
        opts = opts or self.opts
        image = _gray(image)
        h, w = image.shape
        cap, n = 4096, _i()
        c = opts._c()
        while True:
            kl = np.zeros(cap, KEYLINE_DTYPE)
            _lib.check(L.plvs_hip_lsd_detect(self._h, _lib.np_ptr(image), w, h, w, int(numOctaves), float(scale), ctypes.byref(c),
                                             float(opts.min_length), _lib.np_ptr(kl), cap, ctypes.byref(n)))
            if n.value <= cap:
                return kl[:n.value]
            cap = n.value


class LineExtractor:
    skUseLsdExtractor = False

    def __init__(self, numLinefeatures, opts=None):
        opts = opts or LSDOptions()
        self.opts = opts
        self._n = numLinefeatures
        self._lsd = _LsdHandle() if self.skUseLsdExtractor else None      # (read at construction, as the reference's mLsd)
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
        if getattr(self, "_lsd", None):
            self._lsd.close()
            self._lsd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call_lsd(self, image):
        on_device = isinstance(image, torch.Tensor)
        if on_device:
            assert image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2 and image.stride(1) == 1
            torch.cuda.current_stream().synchronize()
            f, ptr, stride = L.plvs_hip_lsd_extract_dev, _vp(image.data_ptr()), image.stride(0)
        else:
            image = _gray(image)
            f, ptr, stride = L.plvs_hip_lsd_extract, _lib.np_ptr(image), image.shape[1]
        h, w = image.shape
        n, c = _i(), self.opts._c()
        while True:
            _lib.check(f(self._lsd._h, ptr, w, h, stride, self._n, self.opts.numOctaves, ctypes.byref(c),
                         float(self.opts.min_length), _lib.np_ptr(self._kl), _lib.np_ptr(self._desc), self._cap, ctypes.byref(n)))
            if n.value <= self._cap:
                break
            self._cap = n.value
            self._kl = np.zeros(self._cap, KEYLINE_DTYPE)
            self._desc = np.zeros((self._cap, 32), np.uint8)
        if n.value == 0:
            print("LineExtractor::detectLineFeatures() - no lines! **********")
        return self._kl[:n.value].copy(), self._desc[:n.value].copy()

    def __call__(self, image):
        """-> (keylines [structured array of KeyLine fields], descriptors [n,32] u8)"""
        if self._lsd is not None:
            return self._call_lsd(image)
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
        if self._lsd is not None:
            # (the reference has no defined result to reproduce here: with the LSD detector BinaryDescriptor::compute takes
            # its sizes from `images_sizes`, which setGaussianPyramid only appends to behind the constructor's zero entries
            # — computeLBD then reads pdxImg[-1]; DESIGN.md §6 "The LSD detector")
            raise NotImplementedError("Line.pyramidPrecomputation together with Line.LSD.on: undefined behaviour in the "
                                      "reference, not reproduced")
        self._shared = orb_extractor                       # keep the handle alive
        L.plvs_hip_lines_set_gaussian_pyramid.argtypes = [_vp, _vp]
        _lib.check(L.plvs_hip_lines_set_gaussian_pyramid(self._h, orb_extractor._h if orb_extractor else None))

    def stage_ms(self):
        if self._lsd is not None:
            return self._lsd.stage_ms()
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
