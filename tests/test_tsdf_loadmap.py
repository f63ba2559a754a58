"""PointCloudMapChisel::LoadMap's integrate (Chisel::IntegrateWorldPointCloudWithNormals, Chisel.cpp:238-376): the
saved map cloud goes back in point by point along its normals.  The oracle's properties on CPU; the HIP path
(ordered pipeline, normals flavour) against the oracle bit for bit through the C ABI."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def surface_cloud(n, seed, res=0.05):
    """Points on the walls of the box |x| <= 3, |y| <= 2, |z| <= 1.5 with inward normals (some un-normalised, a few
    zero or tilted), colours, kfids: what a saved map cloud looks like."""
    rng = np.random.default_rng(seed)
    half = np.array([3.0, 2.0, 1.5])
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    p = rng.uniform(-1, 1, (n, 3)) * half
    p[np.arange(n), axis] = sign * half[axis]
    nrm = np.zeros((n, 3))
    nrm[np.arange(n), axis] = -sign
    nrm += rng.normal(scale=0.05, size=(n, 3))                       # estimated normals are noisy
    nrm *= rng.uniform(0.2, 3.0, (n, 1))                             # and not unit length
    nrm[::97] = 0.0                                                  # degenerate: normalized() leaves a zero vector
    p[::31] = np.round(p[::31] / res) * res                          # points on voxel boundaries
    rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    kfid = rng.integers(0, 50, n).astype(np.uint32)
    return p.astype(np.float32), rgb, kfid, nrm.astype(np.float32)


def chunks_equal(a, b):
    ids = sorted(tuple(int(v) for v in c) for c in a.chunk_ids())
    assert ids == sorted(tuple(int(v) for v in c) for c in b.chunk_ids())
    for cid in ids:
        for name, x, y in zip(("sdf", "weight", "kfid", "colour"), a.get_chunk(*cid), b.get_chunk(*cid)):
            assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y), (name, cid)
    return ids


def test_oracle_world_normals_properties(oracle):
    res = 0.05
    m = oracle.chisel(res)
    xyz, rgb, kfid, nrm = surface_cloud(4000, 1, res)
    m.integrate_world_normals(xyz, rgb, kfid, nrm)
    # each point updates the voxels within 4 voxels of it along its normal: 8-9 of them (1 for a zero normal)
    assert 7.0 * len(xyz) < m.last_visits() < 9.5 * len(xyz)
    trunc = np.float32(4) * np.float32(res)
    wu = np.float32(1.0) / (np.float32(2.0) * trunc)
    seen = 0
    for cid in m.chunk_ids():
        sdf, w, kf, col = m.get_chunk(*cid)
        hit = w > 0
        seen += int(hit.sum())
        assert (np.abs(sdf[hit]) < trunc).all()                          # |u| < truncation for every update
        ratio = w[hit] / wu                                              # constant weight: a whole number of updates
        assert np.allclose(ratio, np.round(ratio), atol=1e-3)
        assert (kf[hit] < 50).all() and ((col[hit] >> 24) >= 1).all()
    assert seen > 5 * len(xyz) // 2
    # a rigid pose moves the map with the cloud: same number of visits for a voxel-aligned translation
    m2 = oracle.chisel(res)
    T = np.eye(4, dtype=np.float32)[:3].copy()
    T[:, 3] = [16 * res, -32 * res, 48 * res]
    m2.integrate_world_normals(xyz, rgb, kfid, nrm, T)
    assert m2.last_visits() == m.last_visits()
    # no depth test: points behind the "camera" integrate too (the camera-ray flavour skips z < 0.01)
    m3 = oracle.chisel(res)
    m3.integrate(xyz, rgb, kfid, np.eye(4, dtype=np.float32)[:3])
    assert m3.last_visits() < m.last_visits()


@pytest.mark.gpu
@pytest.mark.parametrize("res,n,order_free", [(0.05, 30000, False), (0.10, 20000, True), (0.02, 12000, False)])
def test_hip_world_normals_matches_oracle(oracle, res, n, order_free):
    from plvs_amd.tsdf import TsdfChisel
    ref, hip = oracle.chisel(res), TsdfChisel(res, max_chunks=16384, order_free=order_free)
    # onto a map that already holds camera-ray integrations (mixed histories: colour weights of every size)
    for k in make_keyframes(2, seed=4):
        ref.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
        hip.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    T = np.array([[0.0, -1.0, 0.0, 0.3], [1.0, 0.0, 0.0, -0.2], [0.0, 0.0, 1.0, 0.1]], np.float32)
    for seed, pose in ((1, None), (2, T), (3, None)):
        xyz, rgb, kfid, nrm = surface_cloud(n, seed, res)
        ref.integrate_world_normals(xyz, rgb, kfid, nrm, pose)
        hip.integrate_world_normals(xyz, rgb, kfid, nrm, pose)
        assert hip.last_stats()["visits"] == ref.last_visits()
        assert len(hip.updated_chunk_ids()) > 0
    if not order_free:
        chunks_equal(ref, hip)
    else:
        # the handle's camera-ray calls ran order-free (sdf within its stated tolerance); the normals flavour itself is
        # ordered and exact: compare it on a fresh pair of maps
        ref2, hip2 = oracle.chisel(res), TsdfChisel(res, max_chunks=16384, order_free=True)
        xyz, rgb, kfid, nrm = surface_cloud(n, 9, res)
        ref2.integrate_world_normals(xyz, rgb, kfid, nrm)
        hip2.integrate_world_normals(xyz, rgb, kfid, nrm)
        assert len(chunks_equal(ref2, hip2)) > 10
        hip2.close()
    # empty cloud: a no-op
    hip.integrate_world_normals(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), np.zeros(0, np.uint32),
                                np.zeros((0, 3), np.float32))
    assert hip.last_stats()["visits"] == 0
    hip.close()


@pytest.mark.gpu
def test_save_load_round_trip_through_the_mirror(oracle):
    """SaveMap writes the output cloud (UpdateMap's vertices with normals, colours, kfids); LoadMap integrates it
    into an empty map along the normals.  The reloaded surface must sit on the saved one; and the device LoadMap must
    equal the oracle's on the same cloud."""
    from plvs_amd.tsdf import PointCloudMapChisel
    pm = PointCloudMapChisel(0.05)
    for k in make_keyframes(3, seed=6):
        pm.InsertCloud(dict(xyz=k["xyz"], rgb=k["rgb"], kfid=k["kfid"]), k["Twc"])
    saved = pm.UpdateMap()
    assert len(saved) > 8000
    fresh = PointCloudMapChisel(0.05)
    reloaded = fresh.LoadMap(saved)
    ref = oracle.chisel(0.05)
    ref.integrate_world_normals(np.stack([saved["x"], saved["y"], saved["z"]], -1),
                                np.stack([saved["r"], saved["g"], saved["b"]], -1), saved["kfid"], saved["normal"])
    chunks_equal(ref, fresh.tsdf)
    assert len(reloaded) > 0.5 * len(saved)
    # every reloaded vertex lies close to a saved one (the surface did not move): nearest neighbour through a voxel grid
    key = lambda c: set(map(tuple, np.floor(np.stack([c["x"], c["y"], c["z"]], -1) / 0.1).astype(np.int64)))
    cells = key(saved)
    grown = {(a + dx, b + dy, c + dz) for (a, b, c) in cells for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)}
    assert len(key(reloaded) - grown) < 0.01 * len(key(reloaded))


# ---------------------------------------------------------------- voxblox
def test_oracle_voxblox_world_normals_properties(oracle):
    vs, trunc = 0.05, 0.1
    m = oracle.voxblox(vs)
    xyz, rgb, _, nrm = surface_cloud(3000, 5, vs)
    keep = np.linalg.norm(nrm, axis=1) > 0                        # (a zero normal makes 0 / 0 distances: separate case)
    xyz, rgb, nrm = xyz[keep], rgb[keep], nrm[keep]
    rgba = np.concatenate([rgb, np.full((len(rgb), 1), 255, np.uint8)], 1)
    m.integrate_world_normals(xyz, rgba, nrm)
    # a ray of 2 * truncation = 4 voxels: 4 to 8 voxels per point, every one updated with weight 1 (no drop-off in
    # front of the surface; behind it the drop-off scales the weight down)
    assert 3.5 * len(xyz) < m.last_visits() < 8.5 * len(xyz)
    for bid in m.chunk_ids():
        d, w, c = m.get_chunk(*bid)
        hit = w > 0
        assert (np.abs(d[hit]) <= np.float32(trunc)).all()
    # the camera-ray flavour drops points closer than min_ray_length; this one has no such test
    near = (np.linalg.norm(xyz, axis=1) < 2.0).sum()
    assert near > 0


@pytest.mark.gpu
@pytest.mark.parametrize("vs,n", [(0.05, 30000), (0.10, 20000), (0.02, 10000)])
def test_hip_voxblox_world_normals_matches_oracle(oracle, vs, n):
    from plvs_amd.tsdf import TsdfVoxblox
    ref, hip = oracle.voxblox(vs), TsdfVoxblox(vs, max_blocks=65536)
    for k in make_keyframes(2, seed=4):                           # onto a map with camera-ray history
        rgba = np.concatenate([k["rgb"], np.full((len(k["rgb"]), 1), 255, np.uint8)], 1)
        ref.integrate(k["xyz"], rgba, k["Twc"])
        hip.integrate(k["xyz"], rgba, k["Twc"])
    T = np.array([[0.0, -1.0, 0.0, 0.3], [1.0, 0.0, 0.0, -0.2], [0.0, 0.0, 1.0, 0.1]], np.float32)
    for seed, pose in ((1, None), (2, T), (3, None)):
        xyz, rgb, _, nrm = surface_cloud(n, seed, vs)             # zero normals included: NaN distances, handled alike
        rgba = np.concatenate([rgb, np.full((len(rgb), 1), 200, np.uint8)], 1)
        ref.integrate_world_normals(xyz, rgba, nrm, pose)
        hip.integrate_world_normals(xyz, rgba, nrm, pose)
        assert hip.last_stats()["visits"] == ref.last_visits()
    ids = sorted(tuple(int(v) for v in b) for b in ref.chunk_ids())
    assert ids == sorted(tuple(int(v) for v in b) for b in hip.chunk_ids())
    for bid in ids:
        for name, x, y in zip(("distance", "weight", "colour"), ref.get_chunk(*bid), hip.get_chunk(*bid)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
    hip.close()


@pytest.mark.gpu
def test_voxblox_save_load_round_trip_through_the_mirror(oracle):
    from plvs_amd.tsdf import PointCloudMapVoxblox
    pm = PointCloudMapVoxblox(0.05, integration_method="simple")
    for k in make_keyframes(3, seed=6):
        rgba = np.concatenate([k["rgb"], np.full((len(k["rgb"]), 1), 255, np.uint8)], 1)
        pm.InsertCloud(dict(xyz=k["xyz"], rgba=rgba), k["Twc"])
    saved = pm.UpdateMap()
    assert len(saved) > 8000
    fresh = PointCloudMapVoxblox(0.05, integration_method="simple")
    reloaded = fresh.LoadMap(saved)
    ref = oracle.voxblox(0.05)
    ref.integrate_world_normals(np.stack([saved["x"], saved["y"], saved["z"]], -1),
                                np.stack([saved["r"], saved["g"], saved["b"], saved["a"]], -1), saved["normal"])
    ids = sorted(tuple(int(v) for v in b) for b in ref.chunk_ids())
    assert ids == sorted(tuple(int(v) for v in b) for b in fresh.tsdf.chunk_ids())
    for bid in ids:
        for x, y in zip(ref.get_chunk(*bid), fresh.tsdf.get_chunk(*bid)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), bid
    assert len(reloaded) > 0.3 * len(saved)


@pytest.mark.gpu
def test_hip_voxblox_deferred_world_blocks_match_oracle(oracle):
    """plvs_hip_tsdf_voxblox_set_deferred_world_blocks: at every point of camera cloud -> world cloud -> world cloud ->
    camera cloud the device map lists the blocks the oracle lists (pinned to the reference's layer by
    tests/test_oracle_pinned.py), with the same voxels; a waiting block is absent from the updated list and reads as
    missing; switching the flag off publishes what waits."""
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfVoxblox
    vs = 0.05
    ref, hip = oracle.voxblox(vs), TsdfVoxblox(vs, max_blocks=65536)
    ref.set_deferred_world_blocks(True)
    hip.set_deferred_world_blocks(True)

    def same():
        ids = sorted(tuple(int(v) for v in b) for b in ref.chunk_ids())
        assert ids == sorted(tuple(int(v) for v in b) for b in hip.chunk_ids())
        assert hip.num_chunks() == len(ids)
        for bid in ids:
            for name, x, y in zip(("distance", "weight", "colour"), ref.get_chunk(*bid), hip.get_chunk(*bid)):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
        return set(ids)

    kfs = make_keyframes(2, seed=4)
    rgba0 = np.concatenate([kfs[0]["rgb"], np.full((len(kfs[0]["rgb"]), 1), 255, np.uint8)], 1)
    ref.integrate(kfs[0]["xyz"], rgba0, kfs[0]["Twc"])
    hip.integrate(kfs[0]["xyz"], rgba0, kfs[0]["Twc"])
    seen0 = same()
    xyz, rgb, _, nrm = surface_cloud(20000, 1, vs)
    rgba = np.concatenate([rgb, np.full((len(rgb), 1), 200, np.uint8)], 1)
    ref.integrate_world_normals(xyz, rgba, nrm)
    hip.integrate_world_normals(xyz, rgba, nrm)
    assert hip.last_stats()["visits"] == ref.last_visits()
    seen1 = same()
    assert seen1 == seen0                                   # nothing the world cloud created is visible
    upd = set(tuple(int(v) for v in b) for b in hip.updated_chunk_ids())
    assert upd <= seen1                                     # (only blocks of the layer report Block::updated())
    assert hip.last_stats()["new_chunks"] > 0
    xyz2, rgb2, _, nrm2 = surface_cloud(20000, 2, vs)
    rgba2 = np.concatenate([rgb2, np.full((len(rgb2), 1), 200, np.uint8)], 1)
    ref.integrate_world_normals(xyz2, rgba2, nrm2)
    hip.integrate_world_normals(xyz2, rgba2, nrm2)
    assert same() == seen0
    rgba1 = np.concatenate([kfs[1]["rgb"], np.full((len(kfs[1]["rgb"]), 1), 255, np.uint8)], 1)
    ref.integrate(kfs[1]["xyz"], rgba1, kfs[1]["Twc"])
    hip.integrate(kfs[1]["xyz"], rgba1, kfs[1]["Twc"])
    seen3 = same()
    assert len(seen3) > len(seen0) + 10                     # the camera cloud published them, with both clouds' voxels
    # ... and they arrive with their updated() flags: the call's updated list holds them whether or not its rays met them
    upd3 = set(tuple(int(v) for v in b) for b in hip.updated_chunk_ids())
    assert (seen3 - seen0) <= upd3 and upd3 <= seen3
    # the flag off: world-cloud blocks show at once again, and whatever waits is published
    ref.integrate_world_normals(xyz + np.float32(3.0), rgba, nrm)
    hip.integrate_world_normals(xyz + np.float32(3.0), rgba, nrm)
    assert same() == seen3
    ref.set_deferred_world_blocks(False)
    hip.set_deferred_world_blocks(False)
    seen4 = same()
    assert len(seen4) > len(seen3)
    # an EMPTY camera cloud publishes too (integratePointCloud starts with updateLayerWithStoredBlocks)
    ref.set_deferred_world_blocks(True)
    hip.set_deferred_world_blocks(True)
    ref.integrate_world_normals(xyz - np.float32(3.0), rgba, nrm)
    hip.integrate_world_normals(xyz - np.float32(3.0), rgba, nrm)
    assert same() == seen4
    ref.integrate(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[1]["Twc"])
    hip.integrate(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[1]["Twc"])
    seen5 = same()
    assert len(seen5) > len(seen4)
    assert set(tuple(int(v) for v in b) for b in hip.updated_chunk_ids()) == seen5 - seen4
    hip.close()
