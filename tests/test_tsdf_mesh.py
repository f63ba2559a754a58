"""Surface extraction from the chisel map (ChunkManager::RecomputeMesh): the oracle's
properties on CPU, and the HIP path against the oracle, byte for byte, through the C ABI."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def build_map(m, n_kf=4, seed=0):
    kfs = make_keyframes(n_kf, seed=seed)
    for k in kfs:
        m.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    return kfs


def test_oracle_mesh_lies_on_the_scene_surface(oracle):
    m = oracle.chisel(0.05)
    build_map(m, 3)
    ids = m.chunk_ids()
    total, on_wall = 0, 0
    for cid in ids:
        v, n, c, k = m.mesh_chunk(*cid)
        assert len(v) % 3 == 0 and len(v) == len(n) == len(c) == len(k)
        if not len(v):
            continue
        total += len(v)
        # vertices sit inside the chunk's cube of corner-to-corner cells (one voxel of slack for the border cubes)
        lo = np.array(cid, np.float32) * 0.8
        assert (v >= lo - 1e-4).all() and (v <= lo + 0.8 + 0.05 + 1e-4).all()
        ln = np.linalg.norm(n.astype(np.float64), axis=1)
        assert np.allclose(ln[ln > 0], 1.0, atol=1e-5)
        assert (c >= 0).all() and (c <= 1.0 + 1e-6).all()
        assert set(np.unique(k)) <= {0, 1, 2}
        # the room is the box |x|<=3, |y|<=2, |z|<=1.5 (plus three spheres): most vertices lie on a wall
        d = np.minimum.reduce([np.abs(np.abs(v[:, 0]) - 3.0), np.abs(np.abs(v[:, 1]) - 2.0), np.abs(np.abs(v[:, 2]) - 1.5)])
        on_wall += int((d < 0.06).sum())
    assert total > 8000 and on_wall > 0.8 * total
    # a chunk that does not exist has no mesh
    assert len(m.mesh_chunk(1000, 1000, 1000)[0]) == 0


def _neighbourhood(ids):
    s = set()
    for cid in ids:
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    s.add((cid[0] + dx, cid[1] + dy, cid[2] + dz))
    return sorted(s)


@pytest.mark.gpu
@pytest.mark.parametrize("res,n_kf", [(0.05, 4), (0.10, 3)])
def test_hip_mesh_matches_oracle(oracle, res, n_kf):
    from plvs_amd.tsdf import TsdfChisel
    ref, hip = oracle.chisel(res), TsdfChisel(res)
    kfs = make_keyframes(n_kf, seed=1)
    for k in kfs[:-1]:
        ref.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
        hip.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    k = kfs[-1]
    ref.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    hip.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    # Chisel's meshesToUpdate: the 27-neighbourhood of the chunks the last integrate updated (Chisel.cpp:553-568);
    # it contains ids of chunks that do not exist
    todo = np.array(_neighbourhood(map(tuple, hip.updated_chunk_ids())), np.int32)
    existing = set(map(tuple, ref.chunk_ids()))
    assert any(tuple(c) not in existing for c in todo) and any(tuple(c) in existing for c in todo)
    got = hip.mesh_chunks(todo)
    first = got["chunk_first"]
    assert first[0] == 0 and first[-1] == len(got["vertices"]) > 2000
    checked = 0
    for i, cid in enumerate(todo):
        v, n, c, kf = ref.mesh_chunk(*cid)
        a, b = first[i], first[i + 1]
        assert b - a == len(v), (tuple(cid), b - a, len(v))
        if len(v):
            assert got["vertices"][a:b].tobytes() == v.tobytes()
            assert got["kfids"][a:b].tobytes() == kf.tobytes()
            assert got["colors"][a:b].tobytes() == c.tobytes()
            assert got["normals"][a:b].tobytes() == n.tobytes()
            checked += 1
    assert checked > 5
    # empty list, unknown chunks only
    e = hip.mesh_chunks(np.zeros((0, 3), np.int32))
    assert len(e["vertices"]) == 0
    e = hip.mesh_chunks(np.array([[999, 999, 999], [-999, 0, 0]], np.int32))
    assert len(e["vertices"]) == 0 and list(e["chunk_first"]) == [0, 0, 0]


@pytest.mark.gpu
def test_hip_mesh_capacity_is_reported():
    import ctypes
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfChisel
    hip = TsdfChisel(0.05)
    build_map(hip, 2)
    ids = np.ascontiguousarray(hip.chunk_ids(), np.int32)
    first = np.zeros(len(ids) + 1, np.int32)
    n = ctypes.c_int()
    f = _lib.lib.plvs_hip_tsdf_chisel_mesh_chunks
    f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 2
    buf = np.zeros((8, 3), np.float32)
    rc = f(hip._h, _lib.np_ptr(ids), len(ids), _lib.np_ptr(buf), _lib.np_ptr(buf.copy()), _lib.np_ptr(buf.copy()),
           _lib.np_ptr(np.zeros(8, np.uint32)), 8, _lib.np_ptr(first), ctypes.byref(n))
    assert rc == _lib.PLVS_ERR_CAPACITY and n.value > 8 and first[-1] == n.value and (buf == 0).all()
