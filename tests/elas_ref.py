"""Test helper: the reference's libelas pipeline (oracle/_ref/libelas_ref.so = Thirdparty/libelas-gpu/CPU compiled
unmodified, oracle/ref/elas_ref_wrap.cpp) with hooks in the two places its own GPU build overrides
(ElasGPU::computeDisparity / adaptiveMean, Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45).

A hook gets the arguments the reference pipeline hands over; `capture()` records them together with what the reference's
own method returns, `run_with()` lets another implementation (the oracle, the HIP path) fill the results and returns the
pipeline's final disparity maps.  Test infrastructure only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libelas_ref.so")

SUPPORT = np.dtype([("u", np.int32), ("v", np.int32), ("d", np.int32)])
TRIANGLE = np.dtype([("c1", np.int32), ("c2", np.int32), ("c3", np.int32), ("t1a", np.float32), ("t1b", np.float32),
                     ("t1c", np.float32), ("t2a", np.float32), ("t2b", np.float32), ("t2c", np.float32)])
assert SUPPORT.itemsize == 12 and TRIANGLE.itemsize == 36

_DISP = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
_MEAN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
_CAND = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
_LR = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
_MAP = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)


class _Hooks(ctypes.Structure):
    _fields_ = [("compute_disparity", _DISP), ("adaptive_mean", _MEAN), ("user", ctypes.c_void_p), ("support_candidates", _CAND),
                ("left_right_check", _LR), ("remove_small_segments", _MAP), ("gap_interpolation", _MAP)]


def available():
    return os.path.exists(REF)


def _lib():
    lib = ctypes.CDLL(REF)
    lib.ref_elas_process_hooked.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    lib.ref_elas_base_compute_disparity.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.ref_elas_base_adaptive_mean.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return lib


def _view(ptr, dtype, count):
    if count == 0:
        return np.zeros(0, dtype)
    buf = (ctypes.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def _run(left, right, subsampling, plvs, on_disparity, on_mean, on_candidates=None, post=None):
    """on_disparity(call: dict, D: float32 view to fill) / on_mean(elas handle, D view) / on_candidates(dict, D_can int16 view
    [H, W], zeroed); any of them None = the reference's own code."""
    lib = _lib()
    h, w = left.shape
    oh, ow = (h // 2, w // 2) if subsampling else (h, w)
    d1, d2 = np.zeros((oh, ow), np.float32), np.zeros((oh, ow), np.float32)
    errors = []

    def disp(user, call, support, n_support, tri, n_tri, grid, grid_dims, i1, i2, right_image, D):
        try:
            gd = _view(grid_dims, np.int32, 3).copy()
            args = dict(call=call, support=_view(support, SUPPORT, n_support).copy(), tri=_view(tri, TRIANGLE, n_tri).copy(),
                        grid=_view(grid, np.int32, int(gd[0]) * int(gd[1]) * int(gd[2])).copy(), grid_dims=gd,
                        I1_desc=_view(i1, np.uint8, 16 * w * h).copy(), I2_desc=_view(i2, np.uint8, 16 * w * h).copy(),
                        right_image=int(right_image), width=w, height=h, subsampling=int(subsampling), lib=lib)
            on_disparity(args, _view(D, np.float32, oh * ow))
        except Exception as e:  # noqa: BLE001 - re-raised after the C call returns
            errors.append(e)

    def mean(user, elas, D):
        try:
            on_mean(dict(elas=elas, lib=lib, width=w, height=h, subsampling=int(subsampling)), _view(D, np.float32, oh * ow))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def cand(user, i1, i2, D_can, cw, ch):
        try:
            on_candidates(dict(I1_desc=_view(i1, np.uint8, 16 * w * h).copy(), I2_desc=_view(i2, np.uint8, 16 * w * h).copy(),
                               width=w, height=h, subsampling=int(subsampling)), _view(D_can, np.int16, cw * ch).reshape(ch, cw))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    # post: dict with any of 'left_right_check'(D1, D2), 'remove_small_segments'(D), 'gap_interpolation'(D): callables that
    # work IN PLACE on float32 [oh, ow] views
    post = post or {}

    def guarded(f):
        def g(*a):
            try:
                f(*a)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        return g

    def lr(user, D1, D2):
        guarded(post["left_right_check"])(_view(D1, np.float32, oh * ow).reshape(oh, ow), _view(D2, np.float32, oh * ow).reshape(oh, ow))

    def seg(user, D):
        guarded(post["remove_small_segments"])(_view(D, np.float32, oh * ow).reshape(oh, ow))

    def gap(user, D):
        guarded(post["gap_interpolation"])(_view(D, np.float32, oh * ow).reshape(oh, ow))

    hooks = _Hooks(_DISP(disp) if on_disparity else _DISP(), _MEAN(mean) if on_mean else _MEAN(), None,
                   _CAND(cand) if on_candidates else _CAND(), _LR(lr) if "left_right_check" in post else _LR(),
                   _MAP(seg) if "remove_small_segments" in post else _MAP(), _MAP(gap) if "gap_interpolation" in post else _MAP())
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    lib.ref_elas_process_hooked(left.ctypes.data, right.ctypes.data, w, h, w, int(plvs), int(subsampling), d1.ctypes.data,
                                d2.ctypes.data, ctypes.byref(hooks))
    if errors:
        raise errors[0]
    return d1, d2


def reference(left, right, subsampling=False, plvs=True):
    """(D1, D2) of the unhooked reference pipeline."""
    return _run(left, right, subsampling, plvs, None, None)


def capture(left, right, subsampling=False, plvs=True):
    """-> (disparity calls, mean calls, (D1, D2)): every call's inputs and the reference's own result."""
    disp_calls, mean_calls = [], []

    def on_disparity(args, D):
        args["lib"].ref_elas_base_compute_disparity(args["call"], D.ctypes.data)
        args = {k: v for k, v in args.items() if k not in ("call", "lib")}
        args["D"] = D.copy()
        disp_calls.append(args)

    def on_mean(args, D):
        before = D.copy()
        args["lib"].ref_elas_base_adaptive_mean(args["elas"], D.ctypes.data)
        mean_calls.append(dict(D_in=before, D_out=D.copy(), width=args["width"], height=args["height"],
                               subsampling=args["subsampling"]))

    out = _run(left, right, subsampling, plvs, on_disparity, on_mean)
    return disp_calls, mean_calls, out


def run_with(left, right, compute_disparity, adaptive_mean, subsampling=False, plvs=True, support_candidates=None, post=None):
    """The reference pipeline with compute_disparity(args) -> D and adaptive_mean(D_in, width, height, subsampling) -> D
    in ElasGPU's two places (None = the reference's own) and, optionally, support_candidates(args) -> D_can [H, W] int16 as
    the candidate loop of Elas::computeSupportMatches."""
    def on_disparity(args, D):
        D[:] = compute_disparity({k: v for k, v in args.items() if k not in ("call", "lib")}).reshape(-1)

    def on_mean(args, D):
        D[:] = adaptive_mean(D.copy(), args["width"], args["height"], args["subsampling"]).reshape(-1)

    def on_candidates(args, D_can):
        got = support_candidates(args)
        assert got.shape == D_can.shape, (got.shape, D_can.shape)
        D_can[:] = got

    return _run(left, right, subsampling, plvs, on_disparity if compute_disparity else None, on_mean if adaptive_mean else None,
                on_candidates if support_candidates else None, post)
