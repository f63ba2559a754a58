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

    # ------------------------------------------------------------ TSDF
    def chisel(self, resolution, **kw):
        return _ChiselLike(self.lib, "oracle_chisel", resolution, **kw)

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

    def close(self):
        if self.h:
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

    def last_visits(self):
        return int(getattr(self.lib, self.p + "_last_visits")(self.h))

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


HOSTCORE_DIR = os.path.join(ROOT, "tests", "host")
HOSTCORE_SO = os.path.join(HOSTCORE_DIR, "libhostcore.so")


def load_hostcore():
    """Host (g++) build of plvs_amd/csrc/tsdf_chisel_core.hpp — the arithmetic the
    kernels execute — for CPU-side agreement checks against the oracle."""
    src = os.path.join(HOSTCORE_DIR, "tsdf_core_host.cpp")
    hdr = os.path.join(ROOT, "plvs_amd", "csrc", "tsdf_chisel_core.hpp")
    if not os.path.exists(HOSTCORE_SO) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(HOSTCORE_SO):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", HOSTCORE_SO], check=True)
    return ctypes.CDLL(HOSTCORE_SO)


def load():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        build()
    return Oracle(ctypes.CDLL(ORACLE_SO))
