"""Host-side mirror of libelas::ElasGPU (Thirdparty/libelas-gpu/GPU/elas_gpu.h:29-47): the two methods of libelas::Elas the
reference's accelerated build overrides — computeDisparity and adaptiveMean — as PointCloudKeyFrame::ProcessStereoLibelas
reaches them (src/PointCloudKeyFrame.cc:335-432) — and the other device stages of a pair: setImages (the descriptor
images), supportCandidates (the candidate loop of computeSupportMatches), leftRightConsistencyCheck, removeSmallSegments,
gapInterpolation.  Same argument meaning as the reference's methods; the support filters, the triangulation, the planes
and the grid of Elas::process are the caller's.  Preconditions shared with the reference: invalid pixels hold -10,
speckle_sim_threshold < 10; parity is against a reference whose uninitialised reads see zeros (DESIGN §3:
the zero-filling malloc the test build of the reference is compiled with).  The arithmetic runs in libplvs_hip.so; there is no CPU fallback."""
import ctypes

import numpy as np

from . import _lib

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L = _lib.lib

SUPPORT_PT = np.dtype([("u", np.int32), ("v", np.int32), ("d", np.int32)])                     # Elas::support_pt
TRIANGLE = np.dtype([("c1", np.int32), ("c2", np.int32), ("c3", np.int32), ("t1a", np.float32), ("t1b", np.float32),
                     ("t1c", np.float32), ("t2a", np.float32), ("t2b", np.float32), ("t2c", np.float32)])   # Elas::triangle


class _Params(ctypes.Structure):
    _fields_ = [("subsampling", ctypes.c_int32), ("grid_size", ctypes.c_int32), ("match_texture", ctypes.c_int32),
                ("beta", _f), ("gamma", _f), ("sigma", _f), ("sradius", _f),
                ("disp_min", ctypes.c_int32), ("disp_max", ctypes.c_int32), ("candidate_stepsize", ctypes.c_int32),
                ("support_texture", ctypes.c_int32), ("lr_threshold", ctypes.c_int32), ("support_threshold", _f),
                ("speckle_sim_threshold", _f), ("speckle_size", ctypes.c_int32), ("ipol_gap_width", ctypes.c_int32),
                ("add_corners", ctypes.c_int32)]


L.plvs_hip_elas_create.argtypes = [ctypes.POINTER(_Params), ctypes.POINTER(_vp)]
L.plvs_hip_elas_destroy.argtypes = [_vp]
L.plvs_hip_elas_compute_disparity.argtypes = [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
L.plvs_hip_elas_adaptive_mean.argtypes = [_vp, _vp, _i, _i]
L.plvs_hip_elas_support_candidates.argtypes = [_vp, _vp, _vp, _i, _i, _vp]
L.plvs_hip_elas_left_right_check.argtypes = [_vp, _vp, _vp, _i, _i]
L.plvs_hip_elas_set_images.argtypes = [_vp, _vp, _vp, _i, _i, _i]
L.plvs_hip_elas_download_descriptors.argtypes = [_vp, _vp, _vp]
L.plvs_hip_elas_remove_small_segments.argtypes = [_vp, _vp, _i, _i]
L.plvs_hip_elas_gap_interpolation.argtypes = [_vp, _vp, _i, _i]


class ElasGPU:
    class Parameters:
        """The fields of Elas::Parameters (elas.h:62-90) the device stages read; defaults: the ROBOTICS setting (:97-121)
        PLVS starts from, `subsampling` as PLVS sets it (PointCloudMapping::skDownsampleStep even)."""

        def __init__(self, subsampling=False, grid_size=20, match_texture=1, beta=0.02, gamma=3.0, sigma=1.0, sradius=2.0,
                     disp_min=0, disp_max=255, candidate_stepsize=5, support_texture=10, lr_threshold=2, support_threshold=0.85,
                     speckle_sim_threshold=1.0, speckle_size=200, ipol_gap_width=3, add_corners=False):
            self.subsampling, self.grid_size, self.match_texture = bool(subsampling), int(grid_size), int(match_texture)
            self.beta, self.gamma, self.sigma, self.sradius = float(beta), float(gamma), float(sigma), float(sradius)
            self.disp_min, self.disp_max, self.candidate_stepsize = int(disp_min), int(disp_max), int(candidate_stepsize)
            self.support_texture, self.lr_threshold = int(support_texture), int(lr_threshold)
            self.support_threshold = float(support_threshold)
            self.speckle_sim_threshold, self.speckle_size = float(speckle_sim_threshold), int(speckle_size)
            self.ipol_gap_width, self.add_corners = int(ipol_gap_width), bool(add_corners)

    def __init__(self, param=None):
        self.param = param or ElasGPU.Parameters()
        q = self.param
        p = _Params(int(q.subsampling), q.grid_size, q.match_texture, q.beta, q.gamma, q.sigma, q.sradius, q.disp_min, q.disp_max,
                    q.candidate_stepsize, q.support_texture, q.lr_threshold, q.support_threshold, q.speckle_sim_threshold,
                    q.speckle_size, q.ipol_gap_width, int(q.add_corners))
        self._h = _vp()
        _lib.check(L.plvs_hip_elas_create(ctypes.byref(p), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_elas_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out_shape(self, width, height):
        return (height // 2, width // 2) if self.param.subsampling else (height, width)

    def computeDisparity(self, p_support, tri, disparity_grid, grid_dims, I1_desc, I2_desc, right_image, width, height,
                         download=True):
        """p_support: SUPPORT_PT records, tri: TRIANGLE records, disparity_grid / grid_dims as Elas::createGrid leaves them,
        I*_desc: Descriptor::I_desc (16 * width * height bytes; both None = the pair staged by the previous call)
        -> D float32 (-10: no triangle, -1: no match); download=False: the map only stays in HBM (postProcess reads it
        there), None is returned."""
        sup = np.ascontiguousarray(p_support, dtype=SUPPORT_PT)
        tri = np.ascontiguousarray(tri, dtype=TRIANGLE)
        grid = np.ascontiguousarray(disparity_grid, dtype=np.int32)
        gd = np.ascontiguousarray(grid_dims, dtype=np.int32)
        if gd.size != 3 or grid.size < int(gd[0]) * int(gd[1]) * int(gd[2]):
            raise ValueError("disparity_grid is smaller than grid_dims says")
        if (I1_desc is None) != (I2_desc is None):
            raise ValueError("both descriptor images or neither")
        d1 = d2 = None
        if I1_desc is not None:
            d1 = np.ascontiguousarray(I1_desc, dtype=np.uint8)
            d2 = np.ascontiguousarray(I2_desc, dtype=np.uint8)
            if d1.size != 16 * width * height or d2.size != d1.size:
                raise ValueError("a descriptor image has 16 * width * height bytes")
        D = np.empty(self._out_shape(width, height), np.float32) if download else None
        _lib.check(L.plvs_hip_elas_compute_disparity(self._h, _lib.np_ptr(sup), len(sup), _lib.np_ptr(tri), len(tri),
                                                     _lib.np_ptr(grid), _lib.np_ptr(gd), None if d1 is None else _lib.np_ptr(d1),
                                                     None if d2 is None else _lib.np_ptr(d2), int(width), int(height),
                                                     int(bool(right_image)), _lib.np_ptr(D)))
        return D

    def postProcess(self, width, height, postprocess_only_left=True, filter_adaptive_mean=True, download=True):
        """The rest of Elas::process behind its two computeDisparity calls (elas.cpp:100-135) in one call on the maps they
        left in HBM: leftRightConsistencyCheck, removeSmallSegments, gapInterpolation, adaptiveMean — the right map only
        unless postprocess_only_left (PLVS sets it).  -> (D1, D2) float32, or None with download=False (depthDev next)."""
        f = L.plvs_hip_elas_postprocess
        f.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp]
        D1 = np.empty(self._out_shape(width, height), np.float32) if download else None
        D2 = np.empty(self._out_shape(width, height), np.float32) if download else None
        _lib.check(f(self._h, int(width), int(height), int(bool(postprocess_only_left)), int(bool(filter_adaptive_mean)),
                     _lib.np_ptr(D1), _lib.np_ptr(D2)))
        return (D1, D2) if download else None

    def depthDev(self, bf, step, d_depth):
        """PointCloudKeyFrame::ProcessStereoLibelas' disparity -> depth (src/PointCloudKeyFrame.cc:399-420) from the left map
        in HBM into the torch float32 CUDA tensor d_depth [height, width] (what PointCloudGenerator.generate_dev reads)."""
        f = L.plvs_hip_elas_depth_dev
        f.argtypes = [_vp, _f, _i, _vp, _i, _i, _vp]
        assert d_depth.is_cuda and d_depth.is_contiguous() and d_depth.dim() == 2
        _lib.check(f(self._h, float(bf), int(step), _lib.t_ptr(d_depth), d_depth.shape[1], d_depth.shape[0],
                     _lib.current_stream_ptr()))

    def setImages(self, I1, I2):
        """libelas::Descriptor of both images on the device (descriptor.cpp:30-131); they stay staged: supportCandidates and
        computeDisparity then take None for the descriptor images."""
        I1, I2 = np.ascontiguousarray(I1, dtype=np.uint8), np.ascontiguousarray(I2, dtype=np.uint8)
        if I1.ndim != 2 or I1.shape != I2.shape:
            raise ValueError("two u8 images of one size")
        self._img_shape = I1.shape
        _lib.check(L.plvs_hip_elas_set_images(self._h, _lib.np_ptr(I1), _lib.np_ptr(I2), I1.shape[1], I1.shape[0], I1.shape[1]))

    def descriptors(self):
        """The staged descriptor images (parity accessor) -> (I1_desc, I2_desc) uint8 [height * width * 16]."""
        h, w = self._img_shape
        d1, d2 = np.empty(16 * w * h, np.uint8), np.empty(16 * w * h, np.uint8)
        _lib.check(L.plvs_hip_elas_download_descriptors(self._h, _lib.np_ptr(d1), _lib.np_ptr(d2)))
        return d1, d2

    def candidateGrid(self, width, height):
        """(D_can_width, D_can_height, step) of Elas::computeSupportMatches (elas.cpp:420-428)."""
        step = self.param.candidate_stepsize + (self.param.candidate_stepsize % 2 if self.param.subsampling else 0)
        return -(-width // step), -(-height // step), step

    def supportCandidates(self, I1_desc, I2_desc, width, height):
        """The candidate loop of Elas::computeSupportMatches (elas.cpp:434-456) -> D_can int16 [D_can_height, D_can_width]:
        the confirmed disparity of every grid point or -1 (row / column 0: 0).  The reference's filters
        (removeInconsistentSupportPoints, removeRedundantSupportPoints, addCornerSupportPoints) follow on the host.  The
        descriptor images stay staged for the computeDisparity calls of the pair (pass None there)."""
        d1 = d2 = None
        if (I1_desc is None) != (I2_desc is None):
            raise ValueError("both descriptor images or neither (None: the pair setImages staged)")
        if I1_desc is not None:
            d1 = np.ascontiguousarray(I1_desc, dtype=np.uint8)
            d2 = np.ascontiguousarray(I2_desc, dtype=np.uint8)
            if d1.size != 16 * width * height or d2.size != d1.size:
                raise ValueError("a descriptor image has 16 * width * height bytes")
        cw, ch, _ = self.candidateGrid(width, height)
        D_can = np.empty((ch, cw), np.int16)
        _lib.check(L.plvs_hip_elas_support_candidates(self._h, None if d1 is None else _lib.np_ptr(d1),
                                                      None if d2 is None else _lib.np_ptr(d2), int(width), int(height),
                                                      _lib.np_ptr(D_can)))
        return D_can

    def _map(self, D, width, height):
        D = np.ascontiguousarray(D, dtype=np.float32).copy()
        if D.size != int(np.prod(self._out_shape(width, height))):
            raise ValueError("D does not have the size of the disparity map")
        return D.reshape(self._out_shape(width, height))

    def leftRightConsistencyCheck(self, D1, D2, width, height):
        """Elas::leftRightConsistencyCheck (elas.cpp:971-1040) -> the two checked maps (invalid = -10)."""
        D1, D2 = self._map(D1, width, height), self._map(D2, width, height)
        _lib.check(L.plvs_hip_elas_left_right_check(self._h, _lib.np_ptr(D1), _lib.np_ptr(D2), int(width), int(height)))
        return D1, D2

    def removeSmallSegments(self, D, width, height):
        """Elas::removeSmallSegments (elas.cpp:1043-1160) on a map whose invalid pixels are -10."""
        D = self._map(D, width, height)
        _lib.check(L.plvs_hip_elas_remove_small_segments(self._h, _lib.np_ptr(D), int(width), int(height)))
        return D

    def gapInterpolation(self, D, width, height):
        """Elas::gapInterpolation (elas.cpp:1163-1347)."""
        D = self._map(D, width, height)
        _lib.check(L.plvs_hip_elas_gap_interpolation(self._h, _lib.np_ptr(D), int(width), int(height)))
        return D

    def adaptiveMean(self, D, width, height):
        """D (the disparity map of a width x height image; half of it with subsampling) -> the filtered map."""
        D = np.ascontiguousarray(D, dtype=np.float32).copy()
        if D.size != int(np.prod(self._out_shape(width, height))):
            raise ValueError("D does not have the size of the disparity map")
        _lib.check(L.plvs_hip_elas_adaptive_mean(self._h, _lib.np_ptr(D), int(width), int(height)))
        return D
