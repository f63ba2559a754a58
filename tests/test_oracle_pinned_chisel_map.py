"""The chisel oracle pinned END TO END by the reference's own library: oracle/_ref/libchisel_full_ref.so is every source
under Thirdparty/open_chisel/src compiled unmodified from /root/reference against the Eigen stand-in of
oracle/ref/eigen_full (oracle/ref/Makefile; entry points in oracle/ref/chisel_full_ref_wrap.cpp).  The same clouds go
through chisel::Chisel there and through the restatement in oracle/tsdf_chisel.c; every voxel of every chunk (sdf,
weight, kfid, colour, colour weight) and every vertex of every chunk mesh must be identical bit for bit.

The .so is built in the container that has the reference tree and travels with the snapshot; without it these tests
are skipped (nothing here reads /root/reference at run time)."""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_keyframes
from tests.test_tsdf_loadmap import surface_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libchisel_full_ref.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF),
                                reason="oracle/_ref/libchisel_full_ref.so (built where /root/reference exists) not present")

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
TRUNC = (0.0019, -0.00152, 0.001504, 6.0)     # Settings: PointCloudMapping.Chisel truncation (src/PointCloudMapChisel.cc)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class RefChisel:
    """chisel::Chisel + ProjectionIntegrator + PinholeCamera set up as ChiselServer does."""

    def __init__(self, resolution, cam, carving=False, carving_dist=0.05, near=0.05, far=5.0, weight=1.0):
        self.lib = lib = ctypes.CDLL(REF)
        lib.ref_chisel_full_create.restype = _vp
        lib.ref_chisel_full_create.argtypes = [_f] * 6 + [_i, _f] + [_f] * 4 + [_i, _i, _f, _f]
        lib.ref_chisel_full_destroy.argtypes = [_vp]
        lib.ref_chisel_full_integrate.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _vp]
        lib.ref_chisel_full_integrate_world_normals.argtypes = [_vp] * 5 + [_i, _vp]
        lib.ref_chisel_full_num_chunks.argtypes = [_vp]
        lib.ref_chisel_full_chunk_ids.argtypes = [_vp, _vp]
        lib.ref_chisel_full_get_chunk.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]
        lib.ref_chisel_full_update_meshes.argtypes = [_vp]
        lib.ref_chisel_full_mesh_chunk.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i]
        lib.ref_chisel_full_deform.argtypes = [_vp, _vp, _vp, _i]
        self.cam = cam
        self.h = _vp(lib.ref_chisel_full_create(resolution, *TRUNC, weight, int(carving), carving_dist, cam["fx"],
                                                cam["fy"], cam["cx"], cam["cy"], cam["width"], cam["height"], near,
                                                far))

    def close(self):
        if self.h:
            self.lib.ref_chisel_full_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def integrate(self, xyz, rgb, kfid, Twc, depth=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        kfid = np.ascontiguousarray(kfid, np.uint32)
        Twc = np.ascontiguousarray(Twc, np.float32).reshape(3, 4)
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.float32)
            assert depth.shape == (self.cam["height"], self.cam["width"])
        self.lib.ref_chisel_full_integrate(self.h, _ptr(xyz), _ptr(rgb), _ptr(kfid), len(xyz), _ptr(Twc), _ptr(depth))

    def integrate_world_normals(self, xyz, rgb, kfid, normals, Twc=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        kfid = np.ascontiguousarray(kfid, np.uint32)
        normals = np.ascontiguousarray(normals, np.float32)
        Twc = np.ascontiguousarray(np.eye(4, dtype=np.float32)[:3] if Twc is None else Twc, np.float32).reshape(3, 4)
        self.lib.ref_chisel_full_integrate_world_normals(self.h, _ptr(xyz), _ptr(rgb), _ptr(kfid), _ptr(normals),
                                                         len(xyz), _ptr(Twc))

    def num_chunks(self):
        return self.lib.ref_chisel_full_num_chunks(self.h)

    def chunk_ids(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.lib.ref_chisel_full_chunk_ids(self.h, _ptr(ids))
        return ids[:n]

    def get_chunk(self, cx, cy, cz):
        out = (np.empty(4096, np.float32), np.empty(4096, np.float32), np.empty(4096, np.uint32),
               np.empty(4096, np.uint32))
        ok = self.lib.ref_chisel_full_get_chunk(self.h, int(cx), int(cy), int(cz), *[_ptr(a) for a in out])
        return out if ok else None

    def deform(self, kfids, Rt):
        kfids = np.ascontiguousarray(kfids, np.uint32)
        Rt = np.ascontiguousarray(Rt, np.float32).reshape(len(kfids), 12)
        self.lib.ref_chisel_full_deform(self.h, _ptr(kfids), _ptr(Rt), len(kfids))

    def update_meshes(self):
        self.lib.ref_chisel_full_update_meshes(self.h)

    def mesh_chunk(self, cx, cy, cz):
        cap = 4096 * 15
        v, nr, c = (np.zeros((cap, 3), np.float32) for _ in range(3))
        k = np.zeros(cap, np.uint32)
        n = self.lib.ref_chisel_full_mesh_chunk(self.h, int(cx), int(cy), int(cz), _ptr(v), _ptr(nr), _ptr(c), _ptr(k),
                                                cap)
        assert n <= cap
        return v[:n].copy(), nr[:n].copy(), c[:n].copy(), k[:n].copy()


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def small_cam(scale):
    c = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        c[k] = c[k] / scale
    c["width"] //= scale
    c["height"] //= scale
    return c


def maps_identical(ref, ora):
    ir = sorted(tuple(int(v) for v in c) for c in ref.chunk_ids())
    io = sorted(tuple(int(v) for v in c) for c in ora.chunk_ids())
    assert ir == io, f"chunk sets differ: {len(ir)} vs {len(io)}; only ref {sorted(set(ir) - set(io))[:4]}, " \
                     f"only oracle {sorted(set(io) - set(ir))[:4]}"
    for cid in ir:
        for name, a, b in zip(("sdf", "weight", "kfid", "colour"), ref.get_chunk(*cid), ora.get_chunk(*cid)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{name} differs in chunk {cid}"
    return len(ir)


@pytest.mark.parametrize("res,scale,nkf", [(0.05, 4, 5), (0.02, 8, 3), (0.1, 4, 4)])
def test_integrate_point_cloud_equals_the_reference_library(oracle, res, scale, nkf):
    """Chisel::IntegratePointCloudWidthDepth without carving (what PointCloudMapChisel::InsertCloud drives), several
    keyframes into one map: chunk creation, garbage collection and every voxel update."""
    cam = small_cam(scale)
    kfs = make_keyframes(nkf, cam=cam, seed=41)
    ref, ora = RefChisel(res, cam), oracle.chisel(res)
    for kf in kfs:
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        maps_identical(ref, ora)
    assert maps_identical(ref, ora) > 8
    ref.close()


def _depth_image(cam, kind, seed=0):
    h, w = cam["height"], cam["width"]
    rng = np.random.default_rng(seed)
    if kind == "far":
        return np.full((h, w), 4.2, np.float32)
    d = rng.uniform(0.4, 4.8, (h, w)).astype(np.float32)
    d[rng.random((h, w)) < 0.15] = np.nan
    return d


@pytest.mark.parametrize("kind", ["far", "mixed"])
def test_carving_with_the_depth_image_equals_the_reference_library(oracle, kind):
    """The same call with carving on: SetupFrustum, GetChunkIDsIntersecting, CarveWithDepth, then the cloud."""
    cam = small_cam(4)
    kfs = make_keyframes(5, cam=cam, seed=43)
    ref, ora = RefChisel(0.05, cam, carving=True), oracle.chisel(0.05)
    carved_any = False
    for step, kf in enumerate(kfs + kfs[:2]):
        depth = _depth_image(cam, kind, seed=step)
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"], depth=depth)
        n, _ = ora.carve(depth, cam["fx"], cam["fy"], cam["cx"], cam["cy"], kf["Twc"])
        carved_any |= n > 0
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        maps_identical(ref, ora)
    assert carved_any
    ref.close()


def test_world_cloud_with_normals_equals_the_reference_library(oracle):
    """Chisel::IntegrateWorldPointCloudWithNormals (LoadMap), on top of a map built from keyframes and with a pose."""
    cam = small_cam(4)
    ref, ora = RefChisel(0.05, cam), oracle.chisel(0.05)
    for kf in make_keyframes(2, cam=cam, seed=47):
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    xyz, rgb, kfid, nrm = surface_cloud(6000, seed=5)
    Twc = make_keyframes(3, cam=cam, seed=49)[2]["Twc"]
    for T in (None, Twc):
        ref.integrate_world_normals(xyz, rgb, kfid, nrm, T)
        ora.integrate_world_normals(xyz, rgb, kfid, nrm, T)
        maps_identical(ref, ora)
    ref.close()


def test_chunk_meshes_equal_the_reference_library(oracle):
    """Chisel::UpdateMeshes -> ChunkManager::RecomputeMeshes -> RecomputeMesh: vertices, gradient normals, interpolated
    colours and kfids of every chunk the integrates marked."""
    cam = small_cam(4)
    ref, ora = RefChisel(0.05, cam), oracle.chisel(0.05)
    for kf in make_keyframes(4, cam=cam, seed=53):
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    ref.update_meshes()
    total = 0
    for cid in sorted(tuple(int(v) for v in c) for c in ora.chunk_ids()):
        got, want = ora.mesh_chunk(*cid), ref.mesh_chunk(*cid)
        for name, a, b in zip(("vertices", "normals", "colours", "kfids"), got, want):
            assert a.shape == b.shape, f"{name} count differs in chunk {cid}: {a.shape} vs {b.shape}"
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{name} differ in chunk {cid}"
        total += len(want[0])
    assert total > 3000
    ref.close()


def small_motions(kfids, seed, rot=0.03, shift=0.08):
    """One rigid correction per key frame, as a pose-graph optimisation hands them to OnMapChange: a rotation of a
    few degrees and a shift of a few voxels.  -> [n, 12] (R row-major, then t)"""
    rng = np.random.default_rng(seed)
    out = np.zeros((len(kfids), 12), np.float32)
    for i in range(len(kfids)):
        w = rng.normal(scale=rot, size=3)
        th = np.linalg.norm(w)
        k = w / max(th, 1e-12)
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        out[i, :9] = R.astype(np.float32).reshape(9)
        out[i, 9:] = rng.normal(scale=shift, size=3).astype(np.float32)
    return out


@pytest.mark.parametrize("res,drop", [(0.05, False), (0.05, True), (0.1, False)])
def test_deform_equals_the_reference_library(oracle, res, drop):
    """Chisel::Deform after several key frames: the oracle walks the old chunks in the order of a std::unordered_map
    fed with the reference's insert / erase history; colliding voxels then merge in the reference's sequence.  Then
    the map is integrated into and deformed AGAIN (the order of the swapped-in container), and meshed."""
    cam = small_cam(4)
    kfs = make_keyframes(5, cam=cam, seed=61)
    ref, ora = RefChisel(res, cam), oracle.chisel(res).track_order()

    def both(kf):
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.end_call()

    for kf in kfs[:4]:
        both(kf)
    assert [tuple(c) for c in ora.chunk_order()] == [tuple(c) for c in ref.chunk_ids()]   # the container's order itself
    kfids = np.unique(np.concatenate([kf["kfid"] for kf in kfs]))
    if drop:
        kfids = kfids[1:]                                    # voxels of one key frame have no transformation: discarded
    Rt = small_motions(kfids, seed=7)
    n_new, discarded, undefined = ora.deform(kfids, Rt)
    assert undefined == 0                                    # (the reference would index out of bounds there)
    assert (discarded > 0) == drop
    ref.deform(kfids, Rt)
    assert [tuple(c) for c in ora.chunk_order()] == [tuple(c) for c in ref.chunk_ids()]
    assert maps_identical(ref, ora) == n_new > 8
    both(kfs[4])
    both(kfs[0])
    assert [tuple(c) for c in ora.chunk_order()] == [tuple(c) for c in ref.chunk_ids()]
    Rt2 = small_motions(kfids, seed=9, rot=0.2, shift=0.5)   # a large correction: many voxels collide
    _, _, undefined = ora.deform(kfids, Rt2)
    assert undefined == 0
    ref.deform(kfids, Rt2)
    maps_identical(ref, ora)
    ref.close()


def test_deform_moves_the_stored_meshes_as_the_reference_library_does(oracle):
    cam = small_cam(4)
    ref, ora = RefChisel(0.05, cam), oracle.chisel(0.05).track_order()
    for kf in make_keyframes(3, cam=cam, seed=67):
        ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        ora.end_call()
    ref.update_meshes()
    ids = sorted(tuple(int(v) for v in c) for c in ora.chunk_ids())
    before = {cid: ora.mesh_chunk(*cid) for cid in ids}
    kfids = np.array([0, 2], np.uint32)                      # key frame 1's vertices stay where they are
    Rt = small_motions(kfids, seed=11, rot=0.1, shift=0.3)
    ref.deform(kfids, Rt)
    total = moved = 0
    for cid in ids:
        v, nr, col, kf = before[cid]
        want_v, want_n, want_c, want_k = ref.mesh_chunk(*cid)
        got_v, got_n = ora.deform_mesh(v, nr, kf, kfids, Rt)
        assert np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32)), f"vertices differ in chunk {cid}"
        assert np.array_equal(got_n.view(np.uint32), want_n.view(np.uint32)), f"normals differ in chunk {cid}"
        assert np.array_equal(col, want_c) and np.array_equal(kf, want_k)
        total += len(v)
        moved += int((got_v != v).any(axis=1).sum())
    assert 0 < moved < total
    ref.close()
