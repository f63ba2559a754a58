"""Surface extraction from the voxblox map (MeshIntegrator::updateMeshForBlock + getMeshAsPointcloud, SURVEY §8f
row 3): the oracle's properties on CPU, and the HIP path against the oracle, byte for byte, through the C ABI."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes

_vp, _i = ctypes.c_void_p, ctypes.c_int


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def _ptr(a):
    return a.ctypes.data_as(_vp)


def mesh_block(m, bx, by, bz):
    f = m.lib.oracle_voxblox_mesh_block
    f.restype = _i
    f.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _i]
    cap = 4096 * 15
    v, nr = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
    c = np.zeros((cap, 4), np.uint8)
    n = f(m.h, int(bx), int(by), int(bz), _ptr(v), _ptr(nr), _ptr(c), cap)
    assert n <= cap
    return v[:n].copy(), nr[:n].copy(), c[:n].copy()


def set_block(m, bx, by, bz, distance, weight, rgba):
    f = m.lib.oracle_voxblox_set_block
    f.restype = None
    f.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp]
    a = [np.ascontiguousarray(distance, np.float32).reshape(4096), np.ascontiguousarray(weight, np.float32).reshape(4096),
         np.ascontiguousarray(rgba, np.uint32).reshape(4096)]
    f(m.h, int(bx), int(by), int(bz), *[_ptr(x) for x in a])


def rgba_of(kf):
    return np.concatenate([kf["rgb"], np.full((len(kf["rgb"]), 1), 255, np.uint8)], 1)


def test_oracle_mesh_of_an_analytic_sphere(oracle):
    """The meshing restatement on a distance field with a known answer: a sphere of radius 0.55 m sampled into a
    2 x 2 x 2 group of blocks (voxel 0.05).  The part of the surface the blocks can mesh (cubes whose +x/+y/+z
    neighbour block exists) must lie on the sphere, wind outwards, carry flat unit normals and the stored colour."""
    vs, r = 0.05, 0.55
    centre = np.array([0.8, 0.8, 0.8])
    m = oracle.voxblox(vs)
    idx = np.arange(16)
    for bx in range(2):
        for by in range(2):
            for bz in range(2):
                X, Y, Z = np.meshgrid(idx + 16 * bx, idx + 16 * by, idx + 16 * bz, indexing="ij")   # [x, y, z]
                pts = np.stack([X, Y, Z], -1) * vs + vs / 2
                sdf = np.linalg.norm(pts - centre, axis=-1) - r
                sdf_zyx = np.transpose(sdf, (2, 1, 0)).astype(np.float32)       # linear index x + 16 * (y + 16 * z)
                rgba = np.full(4096, 200 | (100 << 8) | (50 << 16) | (255 << 24), np.uint32)
                set_block(m, bx, by, bz, sdf_zyx, np.ones(4096, np.float32), rgba)
    V, N, C = [], [], []
    for bx in range(2):
        for by in range(2):
            for bz in range(2):
                v, n, c = mesh_block(m, bx, by, bz)
                assert len(v) % 3 == 0 and len(v) == len(n) == len(c)
                # vertices sit inside the block's cube of cells (one voxel of slack for the border cubes)
                lo = np.array([bx, by, bz], np.float32) * 0.8
                assert (v >= lo - 1e-4).all() and (v <= lo + 0.8 + vs + 1e-4).all()
                V.append(v); N.append(n); C.append(c)
    v, n, c = np.concatenate(V), np.concatenate(N), np.concatenate(C)
    assert len(v) > 3000
    d = np.linalg.norm(v.astype(np.float64) - centre, axis=1)
    assert np.abs(d - r).max() < 0.15 * vs
    # closed and consistently oriented: the sphere lies inside the 2 x 2 x 2 group, so every cube it crosses is meshed
    key = np.round(v.astype(np.float64) / (vs * 1e-3)).astype(np.int64)
    _, vid = np.unique(key, axis=0, return_inverse=True)
    tri = vid.reshape(-1, 3)
    tri = tri[(tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])]
    fwd = {}
    for a, b in np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]):
        fwd[(a, b)] = fwd.get((a, b), 0) + 1
    assert all(cnt == 1 for cnt in fwd.values()) and all((b, a) in fwd for (a, b) in fwd)
    assert len(np.unique(tri)) - len(fwd) // 2 + len(tri) == 2, "Euler characteristic of a sphere"
    # winding (p1 - p0) x (p2 - p0) = the stored flat normal, pointing outwards (distance grows outwards)
    p = v.reshape(-1, 3, 3).astype(np.float64)
    geo = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    keep = np.linalg.norm(geo, axis=1) > 1e-9
    radial = p.mean(axis=1) - centre
    assert (np.einsum("ij,ij->i", geo[keep], radial[keep]) > 0).all()
    nn = n.reshape(-1, 3, 3)
    assert (nn[:, 0] == nn[:, 1]).all() and (nn[:, 0] == nn[:, 2]).all()
    ln = np.linalg.norm(nn[keep, 0].astype(np.float64), axis=1)
    assert np.allclose(ln, 1.0, atol=1e-5)
    assert (c == np.array([200, 100, 50, 255], np.uint8)).all()
    # a block that does not exist has no mesh; unobserved voxels (weight <= 1e-4) stop a cube
    assert len(mesh_block(m, 9, 9, 9)[0]) == 0
    m2 = oracle.voxblox(vs)
    w = np.ones(4096, np.float32)
    w[::2] = 1e-4
    set_block(m2, 0, 0, 0, np.transpose(sdf, (2, 1, 0)).astype(np.float32), w, np.zeros(4096, np.uint32))
    assert len(mesh_block(m2, 0, 0, 0)[0]) == 0


def test_oracle_cloud_colour_round_trip(oracle):
    """getMeshAsPointcloud sends each channel through c / 255.0 -> float -> * 255.0 -> uint8 (a truncation: one ulp
    below c would lose a level).  In this arithmetic it comes back as the identity for all 256 values; the mirrors
    keep the expression rather than the conclusion."""
    f = oracle.lib.oracle_voxblox_cloud_colour
    f.restype, f.argtypes = ctypes.c_uint8, [ctypes.c_uint8]
    got = np.array([f(c) for c in range(256)], np.uint8)
    want = ((np.arange(256) / 255.0).astype(np.float32).astype(np.float64) * 255.0).astype(np.uint8)
    assert np.array_equal(got, want)
    assert np.array_equal(got, np.arange(256))
    from plvs_amd.tsdf import PointCloudMapVoxblox
    assert np.array_equal(PointCloudMapVoxblox._CLOUD_COLOUR, got)


def test_oracle_mesh_of_an_integrated_map(oracle):
    m = oracle.voxblox(0.05)
    for k in make_keyframes(3, seed=0):
        m.integrate(k["xyz"], rgba_of(k), k["Twc"])
    total, on_wall = 0, 0
    for bid in m.chunk_ids():
        v, n, c = mesh_block(m, *bid)
        total += len(v)
        if len(v):
            d = np.minimum.reduce([np.abs(np.abs(v[:, 0]) - 3.0), np.abs(np.abs(v[:, 1]) - 2.0), np.abs(np.abs(v[:, 2]) - 1.5)])
            on_wall += int((d < 0.06).sum())
    assert total > 8000 and on_wall > 0.8 * total


@pytest.mark.gpu
@pytest.mark.parametrize("vs,n_kf,carving", [(0.05, 4, False), (0.10, 3, True), (0.02, 2, False)])
def test_hip_mesh_matches_oracle(oracle, vs, n_kf, carving):
    from plvs_amd.tsdf import TsdfVoxblox
    ref, hip = oracle.voxblox(vs, carving=carving), TsdfVoxblox(vs, use_carving=carving, max_blocks=65536)
    kfs = make_keyframes(n_kf, seed=1)
    if vs < 0.05:                                   # keep the oracle's share of the test in seconds
        kfs = [dict(k, xyz=k["xyz"][::4], rgb=k["rgb"][::4]) for k in kfs]
    for k in kfs:
        ref.integrate(k["xyz"], rgba_of(k), k["Twc"])
        hip.integrate(k["xyz"], rgba_of(k), k["Twc"])
    ids = sorted(tuple(int(x) for x in b) for b in ref.chunk_ids())
    assert ids == sorted(tuple(int(x) for x in b) for b in hip.chunk_ids())
    # the blocks of the last call, a block that does not exist in the middle of the list, then every block
    upd = [tuple(int(x) for x in b) for b in hip.updated_chunk_ids()]
    assert 0 < len(upd) <= len(ids) and set(upd) <= set(ids)
    for todo in (upd[:5] + [(1000, 1000, 1000)] + upd[5:], ids):
        m = hip.mesh_blocks(np.array(todo, np.int32))
        first = m["block_first"]
        total = 0
        for i, bid in enumerate(todo):
            v, n, c = mesh_block(ref, *bid)
            a, b = int(first[i]), int(first[i + 1])
            assert b - a == len(v), (bid, b - a, len(v))
            assert m["vertices"][a:b].tobytes() == v.tobytes(), bid
            assert m["normals"][a:b].tobytes() == n.tobytes(), bid
            assert m["colors"][a:b].tobytes() == c.tobytes(), bid
            total += len(v)
        assert total == len(m["vertices"]) and total > 3000
    hip.close()


@pytest.mark.gpu
def test_hip_mesh_of_analytic_blocks_and_capacity(oracle):
    """Blocks with hand-made payloads cannot be uploaded through the C ABI (the map is only written by integrate), so
    the analytic check runs on an integrated map: capacity handling and an empty list."""
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfVoxblox
    hip = TsdfVoxblox(0.05)
    k = make_keyframes(1, seed=3)[0]
    hip.integrate(k["xyz"], rgba_of(k), k["Twc"])
    ids = np.ascontiguousarray(hip.chunk_ids(), np.int32)
    assert hip.mesh_blocks(np.zeros((0, 3), np.int32))["vertices"].shape[0] == 0
    f = _lib.lib.plvs_hip_tsdf_voxblox_mesh_blocks
    f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2
    first = np.zeros(len(ids) + 1, np.int32)
    n = ctypes.c_int()
    rc = f(hip._h, _lib.np_ptr(ids), len(ids), None, None, None, 0, _lib.np_ptr(first), ctypes.byref(n))
    assert rc == _lib.PLVS_ERR_CAPACITY and n.value > 1000 and first[-1] == n.value and (np.diff(first) >= 0).all()
    full = hip.mesh_blocks(ids)
    assert len(full["vertices"]) == n.value and np.array_equal(full["block_first"], first)
    hip.close()


@pytest.mark.gpu
def test_update_map_mirror_accumulates_updated_blocks(oracle):
    """PointCloudMapVoxblox.UpdateMap: the blocks of every integrate call since the last UpdateMap are re-meshed,
    the others keep their mesh; the cloud is every block's vertices with the colour round trip."""
    from plvs_amd.tsdf import PointCloudMapVoxblox
    ref, pm = oracle.voxblox(0.05), PointCloudMapVoxblox(0.05, integration_method="simple")
    kfs = make_keyframes(3, seed=2)
    for k in kfs[:2]:
        ref.integrate(k["xyz"], rgba_of(k), k["Twc"])
        pm.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
    cloud = pm.UpdateMap()
    want = {tuple(int(x) for x in b): mesh_block(ref, *b) for b in ref.chunk_ids()}
    assert set(pm.mesh_layer) == set(want)
    stale = dict(want)
    k = kfs[2]
    ref.integrate(k["xyz"], rgba_of(k), k["Twc"])
    pm.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
    pm._flush()                # (the mirror queues insertions: the updated set exists once they are integrated)
    touched = set(pm._updated)
    cloud = pm.UpdateMap()
    n = 0
    for bid in sorted(pm.mesh_layer):
        # a block the third cloud did not touch keeps the mesh of the first UpdateMap — even if a neighbour it reads changed
        v, nr, c = mesh_block(ref, *bid) if bid in touched else stale[bid]
        got = pm.mesh_layer[bid]
        assert got["vertices"].tobytes() == v.tobytes() and got["normals"].tobytes() == nr.tobytes(), bid
        assert got["colors"].tobytes() == c.tobytes(), bid
        k_ = len(v)
        seg = cloud[n:n + k_]
        assert np.array_equal(np.stack([seg["x"], seg["y"], seg["z"]], -1), v)
        assert np.array_equal(seg["normal"], nr)
        lut = PointCloudMapVoxblox._CLOUD_COLOUR
        assert np.array_equal(np.stack([seg["r"], seg["g"], seg["b"]], -1), lut[c[:, :3]])
        n += k_
    assert n == len(cloud) > 5000
    pm.Clear()
    assert len(pm.UpdateMap()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5])
def test_hip_sharded_voxblox_meshes_equal_the_single_device_meshes(oracle, world):
    """A block-hash sharded voxblox map (every rank walks every ray and keeps the blocks it owns): each rank meshes
    ITS blocks after fetching the +x / +y / +z neighbour blocks other ranks own (halo_lookup / export / import, here
    between virtual ranks on one device).  Every block's mesh must be the single-device one byte for byte; the halo
    goes away with the next integrate call and leaves the pool as it was."""
    import torch
    from plvs_amd.shard import owner_of, voxblox_halo_ids
    from plvs_amd.tsdf import TsdfVoxblox
    vs = 0.05
    single = TsdfVoxblox(vs, max_blocks=8192)
    ranks = [TsdfVoxblox(vs, max_blocks=8192, shard_rank=r, shard_count=world) for r in range(world)]
    kfs = make_keyframes(4, seed=2)
    for phase, part in enumerate((kfs[:3], kfs[3:])):
        for k in part:
            for t in [single] + ranks:
                t.integrate(k["xyz"], rgba_of(k), k["Twc"])
        ids = np.array(sorted(tuple(int(v) for v in b) for b in single.chunk_ids()), np.int32)
        own = owner_of(ids, world)
        want = single.mesh_blocks(ids)
        wf = want["block_first"]
        total, moved = 0, 0
        for r, t in enumerate(ranks):
            mine = np.ascontiguousarray(ids[own == r])
            assert sorted(map(tuple, mine.tolist())) == sorted(tuple(int(v) for v in b) for b in t.chunk_ids())
            before = t.num_chunks()
            need = voxblox_halo_ids(mine, world, r)
            assert len(need) and (owner_of(need, world) != r).all()
            for q in range(world):                                   # the exchange, between virtual ranks
                ask = np.ascontiguousarray(need[owner_of(need, world) == q])
                if not len(ask):
                    continue
                d_ids = torch.from_numpy(ask).cuda()
                found = torch.zeros(len(ask), dtype=torch.int32, device="cuda")
                ranks[q].halo_lookup(d_ids, found)
                payload = torch.empty((int(found.sum().item()), t.HALO_WORDS), dtype=torch.int32, device="cuda")
                ranks[q].halo_export(d_ids, found, payload)
                t.halo_import(d_ids, found, payload)
                moved += payload.shape[0]
            got = t.mesh_blocks(mine)
            gf = got["block_first"]
            pos = {tuple(int(v) for v in b): i for i, b in enumerate(ids)}
            for j, b in enumerate(mine):
                i = pos[tuple(int(v) for v in b)]
                a, e, a2, e2 = int(wf[i]), int(wf[i + 1]), int(gf[j]), int(gf[j + 1])
                assert e - a == e2 - a2, (b, e - a, e2 - a2)
                for name, w in (("vertices", 1), ("normals", 1), ("colors", 1)):
                    assert want[name][a:e].tobytes() == got[name][a2:e2].tobytes(), (name, b)
                total += e - a
            assert t.num_chunks() == before, "ghosts are not blocks of the map"
            if phase == 1:
                t.halo_clear()
        assert total == len(want["vertices"]) > 5000 and moved > 0
        # without the halo a border cube that reaches into another rank's block is dropped: the halo matters
        if phase == 1:
            bare = sum(len(t.mesh_blocks(np.ascontiguousarray(ids[own == r]))["vertices"]) for r, t in enumerate(ranks))
            assert bare < total
    # the shards still hold exactly the single-device map
    for r, t in enumerate(ranks):
        for b in t.chunk_ids():
            for x, y in zip(single.get_chunk(*b), t.get_chunk(*b)):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), b
    for t in ranks + [single]:
        t.close()


@pytest.mark.gpu
def test_hip_mesh_of_an_analytic_sphere_through_upload_block(oracle):
    """Blocks uploaded through the C ABI (the device side of TsdfServer::loadMap): the analytic sphere of the oracle
    test, meshed on the device, byte for byte the oracle's mesh — cube configurations an integrated room never
    produces — and a layer saved from one map and loaded into another gives the same meshes."""
    from plvs_amd.tsdf import PointCloudMapVoxblox, TsdfVoxblox
    vs, r = 0.05, 0.55
    centre = np.array([0.8, 0.8, 0.8])
    ref, hip = oracle.voxblox(vs), TsdfVoxblox(vs, max_blocks=64)
    idx = np.arange(16)
    rng = np.random.default_rng(4)
    ids = []
    for bx in range(2):
        for by in range(2):
            for bz in range(2):
                X, Y, Z = np.meshgrid(idx + 16 * bx, idx + 16 * by, idx + 16 * bz, indexing="ij")
                pts = np.stack([X, Y, Z], -1) * vs + vs / 2
                sdf = np.transpose(np.linalg.norm(pts - centre, axis=-1) - r, (2, 1, 0)).astype(np.float32).reshape(-1)
                sdf[rng.integers(0, 4096, 40)] = 0.0                         # exact zeros: "outside" by the >= 0 rule
                w = np.ones(4096, np.float32)
                w[rng.integers(0, 4096, 60)] = np.float32(1e-4)              # unobserved voxels (weight <= min_weight)
                rgba = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
                set_block(ref, bx, by, bz, sdf, w, rgba)
                hip.set_chunk(bx, by, bz, sdf, w, rgba)
                ids.append((bx, by, bz))
    hip.set_chunk(1, 1, 1, *ref.get_chunk(1, 1, 1))                          # replacing a block is idempotent
    assert hip.num_chunks() == 8
    m = hip.mesh_blocks(np.array(ids, np.int32))
    first, total = m["block_first"], 0
    for i, bid in enumerate(ids):
        v, n, c = mesh_block(ref, *bid)
        a, b = int(first[i]), int(first[i + 1])
        assert b - a == len(v) and m["vertices"][a:b].tobytes() == v.tobytes(), bid
        assert m["normals"][a:b].tobytes() == n.tobytes() and m["colors"][a:b].tobytes() == c.tobytes(), bid
        total += len(v)
    assert total > 3000
    # save -> load through the mirror
    src = PointCloudMapVoxblox(vs, max_blocks=64, integration_method="simple")
    for bid in ids:
        src.tsdf.set_chunk(*bid, *hip.get_chunk(*bid))
    layer = src.SaveLayer()
    dst = PointCloudMapVoxblox(vs, max_blocks=64, integration_method="simple")
    assert dst.LoadLayer(layer) and dst.tsdf.num_chunks() == 8
    cloud = dst.UpdateMap()                                                   # every loaded block is marked updated
    assert len(cloud) == total
    got = dst.tsdf.mesh_blocks(np.array(ids, np.int32))
    assert got["vertices"].tobytes() == m["vertices"].tobytes() and got["colors"].tobytes() == m["colors"].tobytes()
    hip.close()
