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
