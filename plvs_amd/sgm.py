"""Host-side mirror of sgm::StereoSGM (Thirdparty/libsgm/include/libsgm.h:57-110) as
PointCloudKeyFrame::ProcessStereoLibsgm uses it (src/PointCloudKeyFrame.cc:435-481): dense disparity of a rectified
pair by semi-global matching.  The arithmetic runs in libplvs_hip.so; there is no CPU fallback."""
import ctypes

import numpy as np

from . import _lib

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L = _lib.lib
L.plvs_hip_sgm_create.argtypes = [_i, _i, _i, _i, _i, _f, ctypes.POINTER(_vp)]
L.plvs_hip_sgm_destroy.argtypes = [_vp]
L.plvs_hip_sgm_execute.argtypes = [_vp, _vp, _vp, _vp]
L.plvs_hip_sgm_execute_dev.argtypes = [_vp, _vp, _vp, _vp, _vp]
L.plvs_hip_sgm_download.argtypes = [_vp, _i, _vp]


class StereoSGM:
    class Parameters:
        def __init__(self, P1=10, P2=120, uniqueness=0.95):
            self.P1, self.P2, self.uniqueness = P1, P2, uniqueness

    def __init__(self, width, height, disparity_size=64, input_depth_bits=8, output_depth_bits=8, param=None):
        if input_depth_bits != 8 or output_depth_bits != 8:
            raise ValueError("depth bits: PLVS feeds 8-bit images and reads an 8-bit disparity; 16 is not built")
        if disparity_size not in (64, 128):
            raise ValueError("disparity size must be 64 or 128")      # std::logic_error in the reference
        param = param or StereoSGM.Parameters()
        self.width, self.height = int(width), int(height)
        self._h = _vp()
        _lib.check(L.plvs_hip_sgm_create(self.width, self.height, disparity_size, param.P1, param.P2, param.uniqueness,
                                         ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.plvs_hip_sgm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def execute(self, left_pixels, right_pixels):
        """-> disparity [height, width] uint8 (0 = invalid)."""
        left = np.ascontiguousarray(left_pixels, dtype=np.uint8)
        right = np.ascontiguousarray(right_pixels, dtype=np.uint8)
        if left.shape != (self.height, self.width) or right.shape != left.shape:
            raise ValueError("image size differs from the one given to the constructor")
        out = np.empty((self.height, self.width), np.uint8)
        _lib.check(L.plvs_hip_sgm_execute(self._h, _lib.np_ptr(left), _lib.np_ptr(right), _lib.np_ptr(out)))
        return out

    def execute_dev(self, d_left, d_right, d_disparity):
        """torch uint8 CUDA tensors [height, width]; asynchronous on the current stream."""
        _lib.check(L.plvs_hip_sgm_execute_dev(self._h, _lib.t_ptr(d_left), _lib.t_ptr(d_right), _lib.t_ptr(d_disparity),
                                              _lib.current_stream_ptr()))

    def stage(self, which):
        """Parity accessor of the last call: 'census_left', 'census_right', 'cost_sum', 'raw_left', 'raw_right',
        'median_left', 'median_right'."""
        code = ["census_left", "census_right", "cost_sum", "raw_left", "raw_right", "median_left", "median_right"].index(which)
        n = self.width * self.height
        out = (np.empty((self.height, self.width), np.uint32) if code < 2 else
               np.empty((self.height, self.width, 64), np.uint16) if code == 2 else
               np.empty((self.height, self.width), np.uint8))
        _lib.check(L.plvs_hip_sgm_download(self._h, code, _lib.np_ptr(out)))
        return out
