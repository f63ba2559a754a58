"""Host-side mirror of the depth -> cloud step feeding the TSDF back ends.

Names follow PointCloudMapping (src/PointCloudMapping.cc): `InitCamGridPoints` builds the
z = 1 back-projection table once per camera (:796-905), `GeneratePointCloudInCameraFrameBGRA`
turns one depth + colour image into the camera-frame cloud (:929-1031).  All arithmetic runs in
libplvs_hip.so; there is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib

POINT_SURFEL = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("kfid", "<u4"),
    ("normal", "<f4", (3,)), ("normal_pad", "<f4"),
    ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
    ("depth", "<f4"), ("label", "<u4"), ("label_confidence", "<u4"),
])  # pcl::PointSurfelSegment, include/PointSurfelSegment.h:63-94
assert POINT_SURFEL.itemsize == 48

_F = ctypes.c_float
_D = ctypes.c_double
_I = ctypes.c_int
_P = ctypes.c_void_p


def _bind():
    lib = _lib.lib
    lib.plvs_hip_cloudgen_num_grid_points.argtypes = [_I, _I, _I]
    lib.plvs_hip_cloudgen_grid_points.argtypes = [_I, _I, _I, _D, _D, _D, _D, _P]
    lib.plvs_hip_cloudgen_create.argtypes = [_I, _I, _I, _P, _P]
    lib.plvs_hip_cloudgen_destroy.argtypes = [_P]
    lib.plvs_hip_cloudgen_generate.argtypes = [_P, _P, _I, _P, _I, _D, _D, ctypes.c_uint32, _P, _I, _P, _P]
    lib.plvs_hip_cloudgen_generate_dev.argtypes = [_P, _P, _I, _P, _I, _D, _D, ctypes.c_uint32] + [_P] * 10
    return lib


def InitCamGridPoints(width, height, step, fx, fy, cx, cy):
    """matCamGridPoints_ for an undistorted camera (mDistCoef[0] == 0), [ngrid, 2] float32."""
    lib = _bind()
    n = lib.plvs_hip_cloudgen_num_grid_points(width, height, step)
    grid = np.empty((n, 2), np.float32)
    _lib.check(lib.plvs_hip_cloudgen_grid_points(width, height, step, fx, fy, cx, cy, _lib.np_ptr(grid)))
    return grid


class PointCloudGenerator:
    """One per camera model (the reference computes the grid table once, :800-801)."""

    def __init__(self, width, height, grid_points, step=2, min_depth=0.01, max_depth=10.0):
        self._lib = _bind()
        self.width, self.height, self.step = int(width), int(height), int(step)
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.ngrid = self._lib.plvs_hip_cloudgen_num_grid_points(self.width, self.height, self.step)
        grid = np.ascontiguousarray(grid_points, dtype=np.float32)
        if grid.ndim != 2 or grid.shape[1] != 2 or grid.shape[0] < self.ngrid:
            raise ValueError("grid_points must be [>= ngrid, 2]")
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.plvs_hip_cloudgen_create(self.width, self.height, self.step, _lib.np_ptr(grid),
                                                      ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plvs_hip_cloudgen_destroy(self._h)
            self._h = None

    __del__ = close

    def GeneratePointCloudInCameraFrameBGRA(self, color, depth, kfid, want_pixel_to_point=True):
        """color: [h, w, 3] u8 (BGR), depth: [h, w] f32 -> (records [n] POINT_SURFEL, pixelToPointIndex)."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        color = np.ascontiguousarray(color, dtype=np.uint8)
        if depth.shape != (self.height, self.width) or color.shape != (self.height, self.width, 3):
            raise ValueError("image size differs from the generator's")
        out = np.zeros(self.ngrid, POINT_SURFEL)
        p2p = np.empty((self.height, self.width), np.int32) if want_pixel_to_point else None
        n = ctypes.c_int()
        _lib.check(self._lib.plvs_hip_cloudgen_generate(
            self._h, _lib.np_ptr(depth), self.width, _lib.np_ptr(color), 3 * self.width, self.min_depth,
            self.max_depth, int(kfid), _lib.np_ptr(out), self.ngrid,
            _lib.np_ptr(p2p) if p2p is not None else None, ctypes.byref(n)))
        return out[:n.value], p2p

    def generate_dev(self, d_color, d_depth, kfid, d_xyz, d_rgb=None, d_rgba=None, d_kfid=None, d_normals=None,
                     d_point_depth=None, d_pixel_to_point=None, sync=True):
        """Device flavour: torch tensors resident in HBM; outputs are caller-provided tensors with room
        for ngrid points.  Returns the number of points (after a stream sync) or None if sync=False."""
        def p(t):
            return _lib.t_ptr(t) if t is not None else None
        n = ctypes.c_int()
        _lib.check(self._lib.plvs_hip_cloudgen_generate_dev(
            self._h, p(d_depth), d_depth.stride(0), p(d_color), d_color.stride(0), self.min_depth, self.max_depth,
            int(kfid), p(d_xyz), p(d_rgb), p(d_rgba), p(d_kfid), p(d_normals), p(d_point_depth),
            p(d_pixel_to_point), None, _lib.current_stream_ptr(), ctypes.byref(n) if sync else None))
        return n.value if sync else None
