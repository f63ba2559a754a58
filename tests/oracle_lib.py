"""ctypes loader of oracle/liboracle.so — the CPU restatement of the reference
hot path.  Test infrastructure: imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_descriptor_distance.argtypes = [_vp, _vp]
        lib.oracle_descriptor_distance.restype = _i
        for f in (lib.oracle_knn2_bf, lib.oracle_knn2_mih):
            f.argtypes = [_vp, _i, _vp, _i, _vp, _vp, _vp]
            f.restype = None

    # ------------------------------------------------------------ Hamming
    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return self.lib.oracle_descriptor_distance(_ptr(a), _ptr(b))

    # ------------------------------------------------------------ ORB
    def orb(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        return OracleOrb(self.lib, nfeatures, scale_factor, nlevels, ini_th, min_th)

    def lines(self, **kw):
        return OracleLines(self.lib, **kw)

    def resize_linear(self, img, dw, dh):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty((dh, dw), np.uint8)
        self.lib.oracle_resize_linear_u8(_ptr(img), img.shape[1], img.shape[0], _ptr(out), dw, dh)
        return out

    def gaussian_blur(self, img, ksize, sigma):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty_like(img)
        self.lib.oracle_gaussian_blur_u8.argtypes = [_vp, _i, _i, _i, ctypes.c_double, _vp]
        self.lib.oracle_gaussian_blur_u8(_ptr(img), img.shape[1], img.shape[0], ksize, sigma, _ptr(out))
        return out

    def gaussian_kernel_q8(self, n, sigma):
        out = np.zeros(n, np.int32)
        self.lib.oracle_gaussian_kernel_q8.argtypes = [_i, ctypes.c_double, _vp]
        self.lib.oracle_gaussian_kernel_q8(n, sigma, _ptr(out))
        return out

    def fast_atan2(self, y, x):
        self.lib.oracle_fast_atan2.restype = _f
        self.lib.oracle_fast_atan2.argtypes = [_f, _f]
        return self.lib.oracle_fast_atan2(y, x)

    def fast(self, img, threshold, nonmax=True):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = img.size
        out = np.zeros((cap, 3), np.float32)
        n = self.lib.oracle_fast(_ptr(img), img.shape[1], img.shape[1], img.shape[0], threshold, int(nonmax), _ptr(out), cap)
        return out[:n]

    # ------------------------------------------------------------ TSDF
    def chisel(self, resolution, **kw):
        return _ChiselLike(self.lib, "oracle_chisel", resolution, **kw)

    def voxblox(self, voxel_size, **kw):
        return _VoxbloxLike(self.lib, "oracle_voxblox", voxel_size, **kw)

    # ------------------------------------------------------------ dense stereo (libsgm)
    def sgm(self, left, right, p1=10, p2=120, uniqueness=0.95, stages=False):
        """-> disparity [h, w] u8, or (disparity, dict of stage arrays) with stages=True."""
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        h, w = left.shape
        disp = np.zeros((h, w), np.uint8)
        st = {}
        if stages:
            st = dict(census_left=np.zeros((h, w), np.uint32), census_right=np.zeros((h, w), np.uint32),
                      cost_sum=np.zeros((h, w, 64), np.uint16), raw_left=np.zeros((h, w), np.uint8),
                      raw_right=np.zeros((h, w), np.uint8), median_left=np.zeros((h, w), np.uint8),
                      median_right=np.zeros((h, w), np.uint8))
        f = self.lib.oracle_sgm
        f.restype = None
        f.argtypes = [_vp, _vp, _i, _i, _i, _i, _f] + [_vp] * 8
        order = ["census_left", "census_right", "cost_sum", "raw_left", "raw_right", "median_left", "median_right"]
        f(_ptr(left), _ptr(right), w, h, p1, p2, uniqueness, _ptr(disp), *[_ptr(st[k]) if stages else None for k in order])
        return (disp, st) if stages else disp

    # ------------------------------------------------------------ libelas: the two methods its GPU build overrides
    ELAS_ROBOTICS = dict(grid_size=20, match_texture=1, beta=0.02, gamma=3.0, sigma=1.0, sradius=2.0)   # elas.h:97-121

    def elas_compute_disparity(self, a, **params):
        """a: the arguments of Elas::computeDisparity as tests/elas_ref.py hands them over -> D float32 [h, w] (halved
        with subsampling)."""
        prm = dict(self.ELAS_ROBOTICS, **params)

        class P(ctypes.Structure):
            _fields_ = [("subsampling", _i), ("grid_size", _i), ("match_texture", _i), ("beta", _f), ("gamma", _f),
                        ("sigma", _f), ("sradius", _f)]
        p = P(int(a["subsampling"]), prm["grid_size"], prm["match_texture"], prm["beta"], prm["gamma"], prm["sigma"],
              prm["sradius"])
        w, h = a["width"], a["height"]
        D = np.zeros((h // 2, w // 2) if a["subsampling"] else (h, w), np.float32)
        f = self.lib.oracle_elas_compute_disparity
        f.restype = None
        f.argtypes = [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
        sup, tri = np.ascontiguousarray(a["support"]), np.ascontiguousarray(a["tri"])
        grid, gd = np.ascontiguousarray(a["grid"], np.int32), np.ascontiguousarray(a["grid_dims"], np.int32)
        f(ctypes.byref(p), _ptr(sup), len(sup), _ptr(tri), len(tri), _ptr(grid), _ptr(gd), _ptr(a["I1_desc"]), _ptr(a["I2_desc"]),
          w, h, int(a["right_image"]), _ptr(D))
        return D

    ELAS_SUPPORT_ROBOTICS = dict(candidate_stepsize=5, disp_min=0, disp_max=255, support_texture=10, lr_threshold=2,
                                 support_threshold=0.85)                                                 # elas.h:97-121

    @staticmethod
    def elas_candidate_grid(width, height, subsampling, candidate_stepsize=5):
        step = candidate_stepsize + (candidate_stepsize % 2 if subsampling else 0)        # elas.cpp:420-428
        return len(range(0, width, step)), len(range(0, height, step)), step

    def elas_support_candidates(self, a, **params):
        """The candidate loop of Elas::computeSupportMatches -> D_can int16 [D_can_height, D_can_width] (row / column 0: 0)."""
        prm = dict(self.ELAS_SUPPORT_ROBOTICS, **params)

        class P(ctypes.Structure):
            _fields_ = [("subsampling", _i), ("candidate_stepsize", _i), ("disp_min", _i), ("disp_max", _i),
                        ("support_texture", _i), ("lr_threshold", _i), ("support_threshold", _f)]
        p = P(int(a["subsampling"]), prm["candidate_stepsize"], prm["disp_min"], prm["disp_max"], prm["support_texture"],
              prm["lr_threshold"], prm["support_threshold"])
        w, h = a["width"], a["height"]
        cw, ch, _ = self.elas_candidate_grid(w, h, a["subsampling"], prm["candidate_stepsize"])
        D_can = np.zeros((ch, cw), np.int16)
        f = self.lib.oracle_elas_support_candidates
        f.restype = None
        f.argtypes = [_vp, _vp, _vp, _i, _i, _vp]
        f(ctypes.byref(p), _ptr(a["I1_desc"]), _ptr(a["I2_desc"]), w, h, _ptr(D_can))
        return D_can

    def elas_descriptor(self, img, half_resolution):
        """libelas::Descriptor of a u8 image -> I_desc uint8 [height * width * 16]."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        desc = np.zeros(16 * w * h, np.uint8)
        f = self.lib.oracle_elas_descriptor
        f.restype = None
        f.argtypes = [_vp, _i, _i, _i, _i, _vp]
        f(_ptr(img), w, h, w, int(half_resolution), _ptr(desc))
        return desc

    # (the three below work IN PLACE on float32 [H, W] disparity maps; defaults: the ROBOTICS setting, elas.h:97-121)
    def elas_left_right_check(self, D1, D2, subsampling, lr_threshold=2):
        f = self.lib.oracle_elas_left_right_check
        f.restype = None
        f.argtypes = [_vp, _vp, _i, _i, _i, _i]
        f(_ptr(D1), _ptr(D2), D1.shape[1], D1.shape[0], int(subsampling), lr_threshold)

    def elas_remove_small_segments(self, D, subsampling, speckle_size=200, speckle_sim_threshold=1.0):
        f = self.lib.oracle_elas_remove_small_segments
        f.restype = None
        f.argtypes = [_vp, _i, _i, _i, _i, _f]
        f(_ptr(D), D.shape[1], D.shape[0], int(subsampling), speckle_size, speckle_sim_threshold)

    def elas_gap_interpolation(self, D, subsampling, ipol_gap_width=3, add_corners=False):
        f = self.lib.oracle_elas_gap_interpolation
        f.restype = None
        f.argtypes = [_vp, _i, _i, _i, _i, _i]
        f(_ptr(D), D.shape[1], D.shape[0], int(subsampling), ipol_gap_width, int(add_corners))

    def elas_adaptive_mean(self, D, width, height, subsampling):
        D = np.ascontiguousarray(D, np.float32).copy()
        f = self.lib.oracle_elas_adaptive_mean
        f.restype = None
        f.argtypes = [_vp, _i, _i, _i]
        f(_ptr(D), width, height, int(subsampling))
        return D

    # ------------------------------------------------------------ stereo (M5)
    def stereo_matches(self, keys_left, desc_left, keys_right, desc_right, pyr_left, pyr_right, scale, inv_scale,
                       mb, mbf):
        """pyr_*: lists of level images (mvImagePyramid).  -> (uRight, depth, score, kept)."""
        kl = np.ascontiguousarray(keys_left, dtype=KP_DTYPE)
        kr = np.ascontiguousarray(keys_right, dtype=KP_DTYPE)
        dl = np.ascontiguousarray(desc_left, dtype=np.uint8)
        dr = np.ascontiguousarray(desc_right, dtype=np.uint8)
        pl = [np.ascontiguousarray(a, dtype=np.uint8) for a in pyr_left]
        pr = [np.ascontiguousarray(a, dtype=np.uint8) for a in pyr_right]
        nl = len(pl)
        lw = np.array([a.shape[1] for a in pl], np.int32)
        lh = np.array([a.shape[0] for a in pl], np.int32)
        assert all(a.shape == b.shape for a, b in zip(pl, pr))
        arr_l = (ctypes.c_void_p * nl)(*[a.ctypes.data for a in pl])
        arr_r = (ctypes.c_void_p * nl)(*[a.ctypes.data for a in pr])
        sc = np.ascontiguousarray(scale, dtype=np.float32)
        isc = np.ascontiguousarray(inv_scale, dtype=np.float32)
        u = np.zeros(max(kl.shape[0], 1), np.float32)
        d = np.zeros(max(kl.shape[0], 1), np.float32)
        score = np.zeros(max(kl.shape[0], 1), np.int32)
        f = self.lib.oracle_stereo_matches
        f.restype = _i
        f.argtypes = [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp]
        kept = f(_ptr(kl), _ptr(dl), kl.shape[0], _ptr(kr), _ptr(dr), kr.shape[0], arr_l, arr_r, _ptr(lw), _ptr(lh), nl,
                 _ptr(sc), _ptr(isc), mb, mbf, _ptr(u), _ptr(d), _ptr(score))
        n = kl.shape[0]
        return u[:n], d[:n], score[:n], kept

    # ------------------------------------------------------------ depth -> cloud (T0)
    def cam_grid_points(self, width, height, step, fx, fy, cx, cy):
        ngrid = ((width + step - 1) // step) * ((height + step - 1) // step)
        grid = np.zeros((ngrid, 2), np.float32)
        d = ctypes.c_double
        self.lib.oracle_cam_grid_points.argtypes = [_i, _i, _i, d, d, d, d, _vp]
        self.lib.oracle_cam_grid_points.restype = None
        self.lib.oracle_cam_grid_points(width, height, step, fx, fy, cx, cy, _ptr(grid))
        return grid

    def cloudgen(self, depth, bgr, grid, step, min_depth, max_depth, kfid, depth_pitch=None, bgr_pitch=None,
                 width=None, height=None):
        """-> (records [n] SURFEL_DTYPE, pixel_to_point [h, w] int32)."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        height = height or depth.shape[0]
        width = width or depth.shape[1]
        depth_pitch = depth_pitch or depth.shape[1]
        bgr_pitch = bgr_pitch or bgr.shape[1] * 3
        grid = np.ascontiguousarray(grid, dtype=np.float32)
        ngrid = ((width + step - 1) // step) * ((height + step - 1) // step)
        out = np.zeros(max(ngrid, 1), SURFEL_DTYPE)
        p2p = np.zeros((height, width), np.int32)
        d = ctypes.c_double
        f = self.lib.oracle_cloudgen
        f.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _vp, d, d, ctypes.c_uint32, _vp, _vp]
        f.restype = _i
        n = f(_ptr(depth), depth_pitch, _ptr(bgr), bgr_pitch, width, height, step, _ptr(grid), min_depth, max_depth,
              kfid, _ptr(out), _ptr(p2p))
        return out[:n], p2p

    def knn2(self, q, t, qmask=None, mih=True):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        idx = np.full((nq, 2), -7, dtype=np.int32)
        dist = np.full((nq, 2), -7, dtype=np.int32)
        if qmask is not None:
            qmask = np.ascontiguousarray(qmask, dtype=np.uint8).reshape(-1)
        f = self.lib.oracle_knn2_mih if mih else self.lib.oracle_knn2_bf
        f(_ptr(q), nq, _ptr(t), nt, _ptr(qmask), _ptr(idx), _ptr(dist))
        return idx, dist


SURFEL_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("kfid", "<u4"), ("normal", "<f4", (3,)),
                         ("normal_pad", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
                         ("depth", "<f4"), ("label", "<u4"), ("label_confidence", "<u4")])

KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])


class OracleOrb:
    def __init__(self, lib, nfeatures, scale_factor, nlevels, ini_th, min_th):
        self.lib = lib
        self.nlevels = nlevels
        lib.oracle_orb_create.restype = _vp
        lib.oracle_orb_create.argtypes = [_i, _f, _i, _i, _i]
        lib.oracle_orb_destroy.argtypes = [_vp]
        lib.oracle_orb_extract.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]
        lib.oracle_orb_features_per_level.argtypes = [_vp, _vp]
        lib.oracle_orb_umax.argtypes = [_vp, _vp]
        lib.oracle_orb_level_size.argtypes = [_vp, _i, _vp, _vp]
        lib.oracle_orb_get_level.argtypes = [_vp, _i, _i, _vp]
        lib.oracle_orb_num_candidates.argtypes = [_vp, _i]
        lib.oracle_orb_get_candidates.argtypes = [_vp, _i, _vp]
        self.h = _vp(lib.oracle_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th))
        self.cap = nfeatures * 2 + 64

    def __del__(self):
        if self.h:
            self.lib.oracle_orb_destroy(self.h)
            self.h = None

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        self.lib.oracle_orb_features_per_level(self.h, _ptr(out))
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        self.lib.oracle_orb_umax(self.h, _ptr(out))
        return out

    def extract(self, img, lap=(0, 0)):
        """-> (monoIndex, keypoints structured array, descriptors [n,32])"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = ctypes.c_int()
        h, w = img.shape if img.ndim == 2 else (0, 0)
        mono = self.lib.oracle_orb_extract(self.h, _ptr(img), w, h, w, lap[0], lap[1], _ptr(kps), _ptr(desc),
                                           self.cap, ctypes.byref(n))
        assert n.value <= self.cap
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level(self, level, blurred=False):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.lib.oracle_orb_level_size(self.h, level, ctypes.byref(w), ctypes.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.lib.oracle_orb_get_level(self.h, level, int(blurred), _ptr(out))
        return out

    def candidates(self, level):
        n = self.lib.oracle_orb_num_candidates(self.h, level)
        out = np.zeros((max(n, 1), 3), np.float32)
        self.lib.oracle_orb_get_candidates(self.h, level, _ptr(out))
        return out[:n]


KEYLINE_DTYPE = np.dtype([("angle", np.float32), ("class_id", np.int32), ("octave", np.int32), ("pt_x", np.float32),
                          ("pt_y", np.float32), ("response", np.float32), ("size", np.float32),
                          ("startPointX", np.float32), ("startPointY", np.float32), ("endPointX", np.float32),
                          ("endPointY", np.float32), ("sPointInOctaveX", np.float32), ("sPointInOctaveY", np.float32),
                          ("ePointInOctaveX", np.float32), ("ePointInOctaveY", np.float32),
                          ("lineLength", np.float32), ("numOfPixels", np.int32)])


class OracleLines:
    def __init__(self, lib, nfeatures=100, nlevels=3, scale=1.2, min_length=0.02, fit_err=1.6):
        self.lib, self.nlevels = lib, nlevels
        lib.oracle_lines_create.restype = _vp
        lib.oracle_lines_create.argtypes = [_i, _i, _f, ctypes.c_double, ctypes.c_double]
        lib.oracle_lines_destroy.argtypes = [_vp]
        lib.oracle_lines_extract.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _i]
        lib.oracle_lines_octave_size.argtypes = [_vp, _i, _vp, _vp]
        lib.oracle_lines_get_map.argtypes = [_vp, _i, _i, _vp]
        lib.oracle_lines_num_in_octave.argtypes = [_vp, _i]
        self.h = _vp(lib.oracle_lines_create(nfeatures, nlevels, scale, min_length, fit_err))
        self.cap = 20000

    def __del__(self):
        if self.h:
            self.lib.oracle_lines_destroy(self.h)
            self.h = None

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        kl = np.zeros(self.cap, KEYLINE_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = self.lib.oracle_lines_extract(self.h, _ptr(img), img.shape[1], img.shape[0], img.shape[1], _ptr(kl),
                                          _ptr(desc), self.cap)
        assert n <= self.cap
        return kl[:n].copy(), desc[:n].copy()

    def set_pyramid(self, levels, num_octaves, scale):
        """LineExtractor::SetGaussianPyramid(pyramid, numOctaves, scale) with border 0; levels = list of
        level images (ORBextractor::mvImagePyramid); [] clears it."""
        lv = [np.ascontiguousarray(a, dtype=np.uint8) for a in levels]
        n = len(lv)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data for a in lv])
        w = np.array([a.shape[1] for a in lv] or [0], np.int32)
        h = np.array([a.shape[0] for a in lv] or [0], np.int32)
        f = self.lib.oracle_lines_set_pyramid
        f.restype = None
        f.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _f]
        f(self.h, ptrs, _ptr(w), _ptr(h), n, num_octaves, scale)

    def octave_size(self, octave):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.lib.oracle_lines_octave_size(self.h, octave, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def octave_map(self, octave, which):
        """which: 'blur' (u8), 'dx', 'dy', 'g' (s16), 'dir' (u8)"""
        w, h = ctypes.c_int(), ctypes.c_int()
        self.lib.oracle_lines_octave_size(self.h, octave, ctypes.byref(w), ctypes.byref(h))
        code = {"blur": 0, "dx": 1, "dy": 2, "g": 3, "dir": 4}[which]
        out = np.zeros((h.value, w.value), np.uint8 if code in (0, 4) else np.int16)
        self.lib.oracle_lines_get_map(self.h, octave, code, _ptr(out))
        return out

    def num_in_octave(self, octave):
        return self.lib.oracle_lines_num_in_octave(self.h, octave)


from tests.pgm import read_pgm  # noqa: E402


def golden(name):
    return read_pgm(os.path.join(ROOT, "tests", "golden", name))


class _ChiselLike:
    """Shared accessor logic for the oracle map and the host build of the device
    arithmetic (same C entry-point shapes, different prefix)."""

    def __init__(self, lib, prefix, resolution, tq=0.0019, tl=-0.00152, tc=0.001504, ts=6.0,
                 weight=1.0, shard_rank=0, shard_count=1):
        self.lib, self.p = lib, prefix
        f = getattr
        f(lib, prefix + "_create").restype = _vp
        f(lib, prefix + "_create").argtypes = [_f] * 6 + [_i, _i]
        f(lib, prefix + "_destroy").argtypes = [_vp]
        f(lib, prefix + "_integrate").argtypes = [_vp, _vp, _vp, _vp, _i, _vp]
        f(lib, prefix + "_last_visits").restype = ctypes.c_longlong
        f(lib, prefix + "_last_visits").argtypes = [_vp]
        f(lib, prefix + "_num_chunks").argtypes = [_vp]
        f(lib, prefix + "_chunk_ids").argtypes = [_vp, _vp]
        f(lib, prefix + "_get_chunk").argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]
        self.h = _vp(f(lib, prefix + "_create")(resolution, tq, tl, tc, ts, weight, shard_rank, shard_count))

    def carve(self, depth, fx, fy, cx, cy, Twc, near=0.05, far=5.0, carving_dist=0.05):
        """-> (number of carved chunks, their ids)  (oracle only)"""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = getattr(self.lib, self.p + "_carve")
        f.restype = _i
        f.argtypes = [_vp, _vp, _i, _i] + [_f] * 6 + [_vp, _f, _vp]
        ids = np.zeros((max(self.num_chunks(), 1), 3), np.int32)
        n = f(self.h, _ptr(depth), depth.shape[1], depth.shape[0], fx, fy, cx, cy, near, far, _ptr(Twc), carving_dist,
              _ptr(ids))
        return n, ids[:n]

    def set_chunk(self, cx, cy, cz, sdf, weight, kfid, rgbw):
        """Test hook: overwrite / create a chunk (4096 voxels, id = (z * 16 + y) * 16 + x)  (oracle only)."""
        f = getattr(self.lib, self.p + "_set_chunk")
        f.restype = None
        f.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]
        a = [np.ascontiguousarray(sdf, np.float32).reshape(4096), np.ascontiguousarray(weight, np.float32).reshape(4096),
             np.ascontiguousarray(kfid, np.uint32).reshape(4096), np.ascontiguousarray(rgbw, np.uint32).reshape(4096)]
        f(self.h, int(cx), int(cy), int(cz), *[_ptr(x) for x in a])

    def mesh_chunk(self, cx, cy, cz):
        """ChunkManager::RecomputeMesh of one chunk -> (vertices, normals, colors [n,3] f32, kfids [n] u32)
        (oracle only)."""
        f = getattr(self.lib, self.p + "_mesh_chunk")
        f.restype = _i
        f.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i]
        cap = 4096 * 15
        v, nr, c = (np.zeros((cap, 3), np.float32) for _ in range(3))
        k = np.zeros(cap, np.uint32)
        n = f(self.h, int(cx), int(cy), int(cz), _ptr(v), _ptr(nr), _ptr(c), _ptr(k), cap)
        assert n <= cap
        return v[:n].copy(), nr[:n].copy(), c[:n].copy(), k[:n].copy()

    def close(self):
        if self.h:
            if getattr(self, "ordered", None):
                self.lib.oracle_chisel_ordered_detach.argtypes = [_vp]
                self.lib.oracle_chisel_ordered_detach(self.ordered)
                self.ordered = None
            getattr(self.lib, self.p + "_destroy")(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def integrate(self, xyz, rgb, kfid, Twc):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        kfid = None if kfid is None else np.ascontiguousarray(kfid, dtype=np.uint32)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        getattr(self.lib, self.p + "_integrate")(self.h, _ptr(xyz), _ptr(rgb), _ptr(kfid), xyz.shape[0], _ptr(Twc))

    def integrate_world_normals(self, xyz, rgb, kfid, normals, Twc=None):
        """Chisel::IntegrateWorldPointCloudWithNormals (oracle only)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        normals = np.ascontiguousarray(normals, dtype=np.float32)
        kfid = None if kfid is None else np.ascontiguousarray(kfid, dtype=np.uint32)
        Twc = np.ascontiguousarray(np.eye(4, dtype=np.float32)[:3] if Twc is None else Twc, dtype=np.float32).reshape(3, 4)
        f = getattr(self.lib, self.p + "_integrate_world_normals")
        f.restype = None
        f.argtypes = [_vp] * 5 + [_i, _vp]
        f(self.h, _ptr(xyz), _ptr(rgb), _ptr(kfid), _ptr(normals), xyz.shape[0], _ptr(Twc))

    def last_visits(self):
        return int(getattr(self.lib, self.p + "_last_visits")(self.h))

    # ---- the exact (double) mean of the same visits beside the reference's f32 running mean (oracle only)
    def track_exact(self, on=True):
        f = getattr(self.lib, self.p + "_track_exact")
        f.restype = None
        f.argtypes = [_vp, _i]
        f(self.h, int(on))

    def get_chunk_exact(self, cx, cy, cz):
        """-> (sdf, weight) [4096] f64: sum(w_u u) / sum(w_u) and sum(w_u) over the visits the map took (0 where unknown)"""
        f = getattr(self.lib, self.p + "_get_chunk_exact")
        f.restype = _i
        f.argtypes = [_vp, _i, _i, _i, _vp, _vp]
        sdf, w = np.zeros(4096, np.float64), np.zeros(4096, np.float64)
        ok = f(self.h, int(cx), int(cy), int(cz), _ptr(sdf), _ptr(w))
        assert ok, "no exact accumulators for this chunk"
        return sdf, w

    # ---- Chisel::Deform (oracle only): the reference's chunk-map order is kept beside the map
    def track_order(self):
        """Attach the std::unordered_map shadow (oracle/tsdf_chisel_deform.cpp) to this (still empty) map; from here on
        call end_call() after every integrate."""
        f = self.lib.oracle_chisel_ordered_attach
        f.restype, f.argtypes = _vp, [_vp]
        self.ordered = _vp(f(self.h))
        self.lib.oracle_chisel_ordered_end_call.argtypes = [_vp]
        self.lib.oracle_chisel_ordered_size.argtypes = [_vp]
        self.lib.oracle_chisel_ordered_order.argtypes = [_vp, _vp]
        self.lib.oracle_chisel_ordered_deform.argtypes = [_vp, _vp, _vp, _i, _vp]
        return self

    def clear(self):
        """Chisel::Reset."""
        self.lib.oracle_chisel_clear.argtypes = [_vp]
        self.lib.oracle_chisel_clear(self.h)
        if getattr(self, "ordered", None):
            self.lib.oracle_chisel_ordered_reset.argtypes = [_vp]
            self.lib.oracle_chisel_ordered_reset(self.ordered)

    def end_call(self):
        self.lib.oracle_chisel_ordered_end_call(self.ordered)

    def chunk_order(self):
        n = self.lib.oracle_chisel_ordered_size(self.ordered)
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.lib.oracle_chisel_ordered_order(self.ordered, _ptr(ids))
        return ids[:n]

    def deform(self, kfids, Rt):
        """kfids [n] strictly increasing, Rt [n, 12] (R row-major, t) -> (new chunk count, discarded, undefined)"""
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        assert np.all(np.diff(kfids.astype(np.int64)) > 0)
        stats = np.zeros(2, np.int64)
        n = self.lib.oracle_chisel_ordered_deform(self.ordered, _ptr(kfids), _ptr(Rt), len(kfids), _ptr(stats))
        return n, int(stats[0]), int(stats[1])

    def deform_mesh(self, vertices, normals, vkfid, kfids, Rt):
        v = np.ascontiguousarray(vertices, np.float32).copy()
        nr = np.ascontiguousarray(normals, np.float32).copy()
        vk = np.ascontiguousarray(vkfid, np.uint32)
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        f = self.lib.oracle_chisel_deform_mesh
        f.restype, f.argtypes = None, [_vp, _vp, _vp, _i, _vp, _vp, _i]
        f(_ptr(v), _ptr(nr), _ptr(vk), len(vk), _ptr(kfids), _ptr(Rt), len(kfids))
        return v, nr

    def num_chunks(self):
        return getattr(self.lib, self.p + "_num_chunks")(self.h)

    def chunk_ids(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), dtype=np.int32)
        getattr(self.lib, self.p + "_chunk_ids")(self.h, _ptr(ids))
        return ids[:n]

    def get_chunk(self, cx, cy, cz):
        sdf = np.empty(4096, np.float32)
        w = np.empty(4096, np.float32)
        kf = np.empty(4096, np.uint32)
        col = np.empty(4096, np.uint32)
        ok = getattr(self.lib, self.p + "_get_chunk")(self.h, int(cx), int(cy), int(cz), _ptr(sdf), _ptr(w), _ptr(kf), _ptr(col))
        return (sdf, w, kf, col) if ok else None


class _VoxbloxLike:
    """Oracle voxblox map / host build of the voxblox device arithmetic."""

    def __init__(self, lib, prefix, voxel_size, truncation=0.1, max_weight=10000.0, min_ray=0.1, max_ray=5.0,
                 carving=False, shard_rank=0, shard_count=1):
        self.lib, self.p = lib, prefix
        g = lambda n: getattr(lib, prefix + n)
        g("_create").restype = _vp
        g("_create").argtypes = [_f] * 5 + [_i, _i, _i]
        g("_destroy").argtypes = [_vp]
        g("_integrate").argtypes = [_vp, _vp, _vp, _i, _vp]
        g("_last_visits").restype = ctypes.c_longlong
        g("_last_visits").argtypes = [_vp]
        g("_num_chunks").argtypes = [_vp]
        g("_chunk_ids").argtypes = [_vp, _vp]
        g("_get_chunk").argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp]
        self.h = _vp(g("_create")(voxel_size, truncation, max_weight, min_ray, max_ray, int(carving), shard_rank, shard_count))

    def __del__(self):
        if self.h:
            getattr(self.lib, self.p + "_destroy")(self.h)
            self.h = None

    def set_deferred_world_blocks(self, enable):
        """Blocks created by integrate_world_normals join the block list only with the next camera-ray integrate."""
        f = getattr(self.lib, self.p + "_set_deferred_world_blocks")
        f.argtypes = [_vp, _i]
        f(self.h, int(enable))

    def integrate(self, xyz, rgba, Twc):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        getattr(self.lib, self.p + "_integrate")(self.h, _ptr(xyz), _ptr(rgba), xyz.shape[0], _ptr(Twc))

    def integrate_fast(self, xyz, rgba, Twc, approx_sets=False):
        """FastTsdfIntegrator::integratePointCloud, one thread (oracle only): with the reference's approximate hash sets
        (approx_sets) or with collision-free sets (what the device builds)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = getattr(self.lib, self.p + "_integrate_fast")
        f.restype = None
        f.argtypes = [_vp, _vp, _vp, _i, _vp, _i]
        f(self.h, _ptr(xyz), _ptr(rgba), xyz.shape[0], _ptr(Twc), int(bool(approx_sets)))

    def integrate_merged(self, xyz, rgba, Twc):
        """MergedTsdfIntegrator::integratePointCloud (oracle only) -> (number of bundles, first point of each)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        Twc = np.ascontiguousarray(Twc, dtype=np.float32).reshape(3, 4)
        f = getattr(self.lib, self.p + "_integrate_merged")
        f.restype = None
        f.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp]
        nb = _i()
        firsts = np.zeros(max(xyz.shape[0], 1), np.int32)
        f(self.h, _ptr(xyz), _ptr(rgba), xyz.shape[0], _ptr(Twc), ctypes.byref(nb), _ptr(firsts))
        return nb.value, firsts[:nb.value]

    def integrate_world_normals(self, xyz, rgba, normals, Twc=None):
        """TsdfIntegratorBase::integrateWorlPointCloud (oracle only)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        normals = np.ascontiguousarray(normals, dtype=np.float32)
        Twc = np.ascontiguousarray(np.eye(4, dtype=np.float32)[:3] if Twc is None else Twc, dtype=np.float32).reshape(3, 4)
        f = getattr(self.lib, self.p + "_integrate_world_normals")
        f.restype = None
        f.argtypes = [_vp] * 4 + [_i, _vp]
        f(self.h, _ptr(xyz), _ptr(rgba), _ptr(normals), xyz.shape[0], _ptr(Twc))

    def last_visits(self):
        return int(getattr(self.lib, self.p + "_last_visits")(self.h))

    # ---- Chisel::Deform (oracle only): the reference's chunk-map order is kept beside the map
    def track_order(self):
        """Attach the std::unordered_map shadow (oracle/tsdf_chisel_deform.cpp) to this (still empty) map; from here on
        call end_call() after every integrate."""
        f = self.lib.oracle_chisel_ordered_attach
        f.restype, f.argtypes = _vp, [_vp]
        self.ordered = _vp(f(self.h))
        self.lib.oracle_chisel_ordered_end_call.argtypes = [_vp]
        self.lib.oracle_chisel_ordered_size.argtypes = [_vp]
        self.lib.oracle_chisel_ordered_order.argtypes = [_vp, _vp]
        self.lib.oracle_chisel_ordered_deform.argtypes = [_vp, _vp, _vp, _i, _vp]
        return self

    def end_call(self):
        self.lib.oracle_chisel_ordered_end_call(self.ordered)

    def chunk_order(self):
        n = self.lib.oracle_chisel_ordered_size(self.ordered)
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.lib.oracle_chisel_ordered_order(self.ordered, _ptr(ids))
        return ids[:n]

    def deform(self, kfids, Rt):
        """kfids [n] strictly increasing, Rt [n, 12] (R row-major, t) -> (new chunk count, discarded, undefined)"""
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        assert np.all(np.diff(kfids.astype(np.int64)) > 0)
        stats = np.zeros(2, np.int64)
        n = self.lib.oracle_chisel_ordered_deform(self.ordered, _ptr(kfids), _ptr(Rt), len(kfids), _ptr(stats))
        return n, int(stats[0]), int(stats[1])

    def deform_mesh(self, vertices, normals, vkfid, kfids, Rt):
        v = np.ascontiguousarray(vertices, np.float32).copy()
        nr = np.ascontiguousarray(normals, np.float32).copy()
        vk = np.ascontiguousarray(vkfid, np.uint32)
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        f = self.lib.oracle_chisel_deform_mesh
        f.restype, f.argtypes = None, [_vp, _vp, _vp, _i, _vp, _vp, _i]
        f(_ptr(v), _ptr(nr), _ptr(vk), len(vk), _ptr(kfids), _ptr(Rt), len(kfids))
        return v, nr

    def num_chunks(self):
        return getattr(self.lib, self.p + "_num_chunks")(self.h)

    def chunk_ids(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), dtype=np.int32)
        getattr(self.lib, self.p + "_chunk_ids")(self.h, _ptr(ids))
        return ids[:n]

    def get_chunk(self, bx, by, bz):
        d = np.empty(4096, np.float32)
        w = np.empty(4096, np.float32)
        c = np.empty(4096, np.uint32)
        ok = getattr(self.lib, self.p + "_get_chunk")(self.h, int(bx), int(by), int(bz), _ptr(d), _ptr(w), _ptr(c))
        return (d, w, c) if ok else None


HOSTCORE_DIR = os.path.join(ROOT, "tests", "host")


ROCM_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _host_build(stem, src, hdrs):
    """Builds a host harness on demand.  PLVS_HOST_CXX=rocm-clang compiles it with the compiler and
    optimisation level the product's host code is built with (hipcc's clang, -O3) instead of g++ -O2."""
    if os.environ.get("PLVS_HOST_CXX") == "rocm-clang":
        cxx, opt, so = ROCM_CLANG, "-O3", os.path.join(HOSTCORE_DIR, stem + "_clang.so")
    else:
        cxx, opt, so = "g++", "-O2", os.path.join(HOSTCORE_DIR, stem + ".so")
    if not os.path.exists(so) or max(os.path.getmtime(f) for f in [src] + hdrs) > os.path.getmtime(so):
        subprocess.run([cxx, opt, "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so], check=True)
    return so


def load_hostcore():
    """Host (g++) build of plvs_amd/csrc/tsdf_chisel_core.hpp — the arithmetic the
    kernels execute — for CPU-side agreement checks against the oracle."""
    src = os.path.join(HOSTCORE_DIR, "tsdf_core_host.cpp")
    hdr = os.path.join(ROOT, "plvs_amd", "csrc", "tsdf_chisel_core.hpp")
    hdr2 = os.path.join(ROOT, "plvs_amd", "csrc", "tsdf_voxblox_core.hpp")
    return ctypes.CDLL(_host_build("libhostcore", src, [hdr, hdr2]))


def load_hostorb():
    """Host (g++) build of plvs_amd/csrc/orb_octree.hpp (the product's quadtree)."""
    src = os.path.join(HOSTCORE_DIR, "orb_host.cpp")
    hdr = os.path.join(ROOT, "plvs_amd", "csrc", "orb_octree.hpp")
    lib = ctypes.CDLL(_host_build("libhostorb", src, [hdr]))
    lib.hostorb_distribute.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _i]
    return lib


def load_hostlines():
    """Host (g++) build of plvs_amd/csrc/lines_host.hpp (the product's sequential line stages)."""
    src = os.path.join(HOSTCORE_DIR, "lines_host.cpp")
    hdr = os.path.join(ROOT, "plvs_amd", "csrc", "lines_host.hpp")
    return ctypes.CDLL(_host_build("libhostlines", src, [hdr]))


def load():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
            if f.endswith((".c", ".h", ".cpp", ".hpp", ".inc"))]
    if not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        build()
    return Oracle(ctypes.CDLL(ORACLE_SO))
