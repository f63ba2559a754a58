"""Points + lines of one frame, extracted concurrently (Frame::Frame's threadLeft /
threadLines, reference src/Frame.cc:503-508).  All compute runs in libplvs_hip.so."""
import ctypes

import torch

from . import _lib
from .lines import LineExtractor
from .orb import ORBextractor

_vp, _i = ctypes.c_void_p, ctypes.c_int
L = _lib.lib
L.plvs_hip_frame_extract_dev.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp,
                                         _vp, _vp, _i, _vp]


_HOOK = ctypes.CFUNCTYPE(None, _vp, _i)
L.plvs_hip_frame_extract_dev_hook.argtypes = L.plvs_hip_frame_extract_dev.argtypes + [_HOOK, _vp]


def extract_frame(orb: ORBextractor, lines: LineExtractor, image: torch.Tensor, vLappingArea=(0, 0), after_points=None):
    """image: 2-D uint8 CUDA tensor.  -> (monoIndex, keypoints, descriptors, keylines, line descriptors)
    — and, with after_points, a sixth element: what the hook returned (None when it did not run).

    after_points(keypoints, descriptors): called on this thread as soon as the points are out, while the line thread
    is still extracting (what the caller does with the points alone: the ORB SearchByProjection of the tracking
    step)."""
    assert image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2
    torch.cuda.current_stream().synchronize()
    h, w = image.shape
    nk, mono, nl = _i(), _i(), _i()
    args = (orb._h, lines._h, _vp(image.data_ptr()), w, h, image.stride(0), vLappingArea[0], vLappingArea[1],
            _lib.np_ptr(orb._kps), _lib.np_ptr(orb._desc), orb._cap, ctypes.byref(nk), ctypes.byref(mono),
            _lib.np_ptr(lines._kl), _lib.np_ptr(lines._desc), lines._cap, ctypes.byref(nl))
    if after_points is None:
        _lib.check(L.plvs_hip_frame_extract_dev(*args))
    else:
        failure, hooked = [], [None]

        def hook(_user, status):
            if status != 0 or nk.value > orb._cap:
                return
            try:
                hooked[0] = after_points(orb._kps[:nk.value], orb._desc[:nk.value])
            except BaseException as e:      # (an exception must not unwind through the C frame)
                failure.append(e)

        _lib.check(L.plvs_hip_frame_extract_dev_hook(*args, _HOOK(hook), None))
        if failure:
            raise failure[0]
    if nk.value > orb._cap or nl.value > lines._cap:
        raise RuntimeError("extract_frame: output capacity exceeded")
    out = (mono.value, orb._kps[:nk.value].copy(), orb._desc[:nk.value].copy(),
           lines._kl[:nl.value].copy(), lines._desc[:nl.value].copy())
    return out if after_points is None else out + (hooked[0],)


# ---------------------------------------------------------------------------------------------- Frame glue
# What Frame::Frame runs between ExtractORB / ExtractLSD and the first search (src/Frame.cc:541-580).  K = (fx, fy, cx, cy);
# dist = mDistCoef (4, 5 or 8 coefficients; None / first coefficient 0: no distortion).
def _calib(K, dist):
    import numpy as np
    K4 = np.ascontiguousarray(K, np.float32).reshape(4)
    d = None if dist is None else np.ascontiguousarray(dist, np.float32).reshape(-1)
    return K4, d, 0 if d is None else int(d.shape[0])


def UndistortKeyPoints(kps, K, dist):
    """Frame::UndistortKeyPoints (src/Frame.cc:1507-1552): mvKeys -> mvKeysUn (KP_DTYPE records)."""
    import numpy as np
    from .orb import KP_DTYPE
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    un = np.empty_like(kps)
    K4, d, nd = _calib(K, dist)
    f = L.plvs_hip_frame_undistort_keypoints
    f.argtypes = [_vp, _i, _vp, _vp, _i, _vp]
    _lib.check(f(_lib.np_ptr(kps), len(kps), _lib.np_ptr(K4), _lib.np_ptr(d), nd, _lib.np_ptr(un)))
    return un


def ComputeImageBounds(width, height, K, dist):
    """Frame::ComputeImageBounds (src/Frame.cc:1749-1778) -> (mnMinX, mnMaxX, mnMinY, mnMaxY, mnMaxDiag)."""
    import numpy as np
    b = np.zeros(5, np.float32)
    K4, d, nd = _calib(K, dist)
    f = L.plvs_hip_frame_compute_image_bounds
    f.argtypes = [_i, _i, _vp, _vp, _i, _vp]
    _lib.check(f(int(width), int(height), _lib.np_ptr(K4), _lib.np_ptr(d), nd, _lib.np_ptr(b)))
    return tuple(float(x) for x in b)


def UndistortKeyLines(keylines, K, dist, bounds):
    """Frame::UndistortKeyLines (src/Frame.cc:1555-1700, single pinhole camera) -> (mvKeyLinesUn, kept): the undistorted
    lines that stay inside bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY) and the indices of the input lines behind them (the
    caller compacts mvKeyLines / mLineDescriptors with `kept`)."""
    import numpy as np
    kl = np.ascontiguousarray(keylines)
    assert kl.dtype.itemsize == 68
    un = np.empty_like(kl)
    kept = np.zeros(max(len(kl), 1), np.int32)
    n = _i()
    K4, d, nd = _calib(K, dist)
    b = np.ascontiguousarray(bounds, np.float32)[:4].copy()
    f = L.plvs_hip_frame_undistort_keylines
    f.argtypes = [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]
    _lib.check(f(_lib.np_ptr(kl), len(kl), _lib.np_ptr(K4), _lib.np_ptr(d), nd, _lib.np_ptr(b), _lib.np_ptr(un), _lib.np_ptr(kept),
                 ctypes.byref(n)))
    return un[:n.value].copy(), kept[:n.value].copy()


def AssignFeaturesToGrid(kps_un, min_x, min_y, grid_w_inv, grid_h_inv):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:716-746), the key-point grid as a CSR -> (cell_start [3073], cell_items):
    cell = column * 48 + row (mGrid[ix][iy]), members in key-point order."""
    import numpy as np
    from .orb import KP_DTYPE
    kps = np.ascontiguousarray(kps_un, KP_DTYPE)
    start = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(len(kps), 1), np.int32)
    n = _i()
    f = L.plvs_hip_frame_assign_features_to_grid
    f.argtypes = [_vp, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]
    _lib.check(f(_lib.np_ptr(kps), len(kps), float(min_x), float(min_y), float(grid_w_inv), float(grid_h_inv), _lib.np_ptr(start),
                 _lib.np_ptr(items), ctypes.byref(n)))
    return start, items[:n.value].copy()
