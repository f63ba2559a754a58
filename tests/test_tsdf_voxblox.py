"""Voxblox TSDF integrate ("simple" integrator, single-thread order): oracle
self-checks + device arithmetic on the host (CPU), HIP path vs oracle (GPU).

Bar: BIT-EXACT against the oracle for distance, weight and the u8 colours
(TOL = 0; the north-star only asks for a float tolerance).
"""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes, TUM1


def small_cam(scale=4):
    c = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        c[k] = c[k] / scale
    c["width"] //= scale
    c["height"] //= scale
    return c


def rgba_of(kf):
    return np.concatenate([kf["rgb"], np.full((kf["rgb"].shape[0], 1), 255, np.uint8)], axis=1)


def compare(a, b):
    ia, ib = {tuple(x) for x in a.chunk_ids()}, {tuple(x) for x in b.chunk_ids()}
    assert ia == ib, f"block sets differ: {len(ia)} vs {len(ib)}"
    for bid in sorted(ia):
        ca, cb = a.get_chunk(*bid), b.get_chunk(*bid)
        assert np.array_equal(ca[0].view(np.uint32), cb[0].view(np.uint32)), f"distance differs in block {bid}"
        assert np.array_equal(ca[1].view(np.uint32), cb[1].view(np.uint32)), f"weight differs in block {bid}"
        assert np.array_equal(ca[2], cb[2]), f"colour differs in block {bid}"
    return len(ia)


# ------------------------------------------------------------------ CPU
def test_oracle_single_point_known_answer(oracle):
    """One slightly off-axis point 1 m ahead, identity pose, 5 cm voxels, no
    carving: the ray covers +-0.1 m around the point; every voxel on it holds
    sdf = |p| - <c, p>/|p| (clamped to +-0.1) and weight 1/z^2, scaled by the
    linear drop-off (0.1 + sdf)/(0.1 - 0.05) behind the surface."""
    m = oracle.voxblox(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    p = np.array([0.011, 0.017, 1.0])
    m.integrate(p[None].astype(np.float32), np.array([[10, 20, 30, 255]], np.uint8), Twc)
    assert m.last_visits() in (4, 5)     # voxels z = 18..21(22): 0.9 .. 1.1 m along the ray
    hit = 0
    for bid in m.chunk_ids():
        d, w, c = m.get_chunk(*bid)
        for vid in np.nonzero(w > 0)[0]:
            g = np.array([bid[0] * 16 + (vid & 15), bid[1] * 16 + ((vid >> 4) & 15), bid[2] * 16 + (vid >> 8)])
            ctr = (g + 0.5) * 0.05
            sdf = np.linalg.norm(p) - ctr.dot(p) / np.linalg.norm(p)
            wexp = 1.0 if sdf >= -0.05 else max((0.1 + sdf) / 0.05, 0.0)
            assert abs(d[vid] - np.clip(sdf, -0.1, 0.1)) < 1e-5
            assert abs(w[vid] - wexp) < 1e-4
            assert c[vid] == (10 | 20 << 8 | 30 << 16 | 255 << 24)
            hit += 1
    assert hit in (3, 4, 5)              # the last voxel can sit at weight 0 (full drop-off)


def test_oracle_axis_aligned_ray_quirk(oracle):
    """A ray with an exactly zero component that starts on a voxel face makes
    RayCaster's t_to_next_boundary NaN in that axis; Eigen's minCoeff then keeps
    picking it, the caster never advances and the first voxel is updated
    steps+1 times.  The restatement (and the device path) keep that behaviour."""
    m = oracle.voxblox(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    m.integrate(np.array([[0.0, 0.0, 1.0]], np.float32), np.array([[10, 20, 30, 255]], np.uint8), Twc)
    assert m.last_visits() == 5 and m.num_chunks() == 1
    d, w, c = m.get_chunk(*m.chunk_ids()[0])
    assert np.count_nonzero(w) == 1 and w.max() == 5.0


def test_oracle_range_gating_and_mixed_order(oracle):
    m = oracle.voxblox(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    pts = np.array([[0.003, 0.002, 0.05], [0.013, 0.021, 6.0], [0.007, 0.009, 2.0]], np.float32)  # too close, too far, ok
    m.integrate(pts, np.zeros((3, 4), np.uint8), Twc)
    assert m.last_visits() in (4, 5, 6)              # only the 2 m point: ~0.2 m of ray
    carve = oracle.voxblox(0.05, carving=True)
    carve.integrate(pts, np.zeros((3, 4), np.uint8), Twc)
    # carving: the valid ray starts at the origin (~42 voxels to 2.1 m) and the far
    # point becomes a clearing ray of min(6 - 0.1, 5) = 5 m (~100 voxels)
    assert 40 + 98 <= carve.last_visits() <= 46 + 104


@pytest.mark.parametrize("res,carving", [(0.05, False), (0.02, False), (0.10, True)])
def test_device_arithmetic_on_host_matches_oracle(oracle, res, carving):
    host = oracle_lib._VoxbloxLike(oracle_lib.load_hostcore(), "hostvbx", res, carving=carving)
    ora = oracle.voxblox(res, carving=carving)
    for kf in make_keyframes(3, cam=small_cam(4), seed=3):
        ora.integrate(kf["xyz"], rgba_of(kf), kf["Twc"])
        host.integrate(kf["xyz"], rgba_of(kf), kf["Twc"])
        assert ora.last_visits() == host.last_visits() > 0
    assert compare(ora, host) > 3


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("res,nkf,scale,carving", [(0.05, 3, 4, False), (0.02, 2, 4, False), (0.10, 2, 2, True),
                                                   (0.05, 2, 1, False)])
def test_hip_matches_oracle_bit_exact(oracle, res, nkf, scale, carving):
    from plvs_amd.tsdf import TsdfVoxblox
    ora = oracle.voxblox(res, carving=carving)
    dev = TsdfVoxblox(res, use_carving=carving, max_blocks=8192)
    for kf in make_keyframes(nkf, cam=small_cam(scale), seed=21):
        ora.integrate(kf["xyz"], rgba_of(kf), kf["Twc"])
        dev.integrate(kf["xyz"], rgba_of(kf), kf["Twc"])
        assert dev.last_stats()["visits"] == ora.last_visits()
    assert compare(ora, dev) > 3
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("max_weight", [10000.0, 2.5, 0.3])
def test_hip_long_voxel_runs_weight_ceiling_and_weight_floor_match_oracle(oracle, max_weight):
    """The fold's eight-lane path (voxel runs of 16 visits and more) and its exceptions: many key frames from one pose
    in ONE call give every surface voxel a run of dozens to hundreds of visits, some running past the fold's 2048-record
    chunks; a small max_weight makes updateTsdfVoxel's weight ceiling act inside nearly every trip of eight visits; points
    two kilometres away (clearing rays of weight 1/z^2 = 2.5e-7) meet its 1e-6 floor on fresh voxels."""
    import torch
    from plvs_amd.tsdf import TsdfVoxblox
    kfs = make_keyframes(2, cam=small_cam(2), seed=29)
    kfs = [kfs[0]] * 5 + [kfs[1]] * 4                      # the same views again and again: long runs, in order
    far = kfs[0]["xyz"][::7].copy()
    far *= (2000.0 / np.maximum(np.abs(far[:, 2:3]), 1e-3))   # the same directions, z = 2 km
    clouds = [np.concatenate([k["xyz"], far]) if i % 3 == 0 else k["xyz"] for i, k in enumerate(kfs)]
    cols = [np.concatenate([rgba_of(k), rgba_of(kfs[0])[::7]]) if i % 3 == 0 else rgba_of(k) for i, k in enumerate(kfs)]
    ora = oracle.voxblox(0.10, carving=True, max_weight=max_weight)
    for c, col, k in zip(clouds, cols, kfs):
        ora.integrate(c, col, k["Twc"])
    dev = TsdfVoxblox(0.10, use_carving=True, max_blocks=8192, max_weight=max_weight)
    xyz = torch.from_numpy(np.concatenate(clouds)).cuda()
    rgba = torch.from_numpy(np.concatenate(cols)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [c.shape[0] for c in clouds]).astype(np.int32)
    dev.integrate_batch_dev(xyz, rgba, offsets, Twc)
    torch.cuda.synchronize()
    st = dev.last_stats()
    assert st["max_run"] > 2048 // 4 and st["visits"] > 40 * 2048      # runs of hundreds of visits, dozens of chunks
    assert compare(ora, dev) > 3
    # and a second call on the populated map (voxels that start from their stored state)
    ora.integrate(clouds[0], cols[0], kfs[0]["Twc"])
    dev.integrate(clouds[0], cols[0], kfs[0]["Twc"])
    compare(ora, dev)
    dev.close()


@pytest.mark.gpu
def test_hip_batch_shards_and_errors(oracle):
    import torch
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfVoxblox
    kfs = make_keyframes(4, cam=small_cam(2), seed=23)
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgba = torch.from_numpy(np.concatenate([rgba_of(k) for k in kfs])).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    for rank, count in ((0, 1), (0, 2), (1, 2), (2, 3)):
        ora = oracle.voxblox(0.05, shard_rank=rank, shard_count=count)
        for k in kfs:
            ora.integrate(k["xyz"], rgba_of(k), k["Twc"])
        dev = TsdfVoxblox(0.05, max_blocks=8192, shard_rank=rank, shard_count=count)
        dev.integrate_batch_dev(xyz, rgba, offsets, Twc)
        torch.cuda.synchronize()
        compare(ora, dev)
        dev.close()
    dev = TsdfVoxblox(0.05, max_blocks=8192)
    bad = kfs[0]["xyz"].copy()
    bad[5, 1] = np.nan
    with pytest.raises(_lib.PlvsHipError) as e:
        dev.integrate(bad, rgba_of(kfs[0]), kfs[0]["Twc"])
    assert e.value.code == _lib.PLVS_ERR_INVALID_ARG
    dev.close()


def test_oracle_pose_goes_through_the_kindr_quaternion():
    """T_G_C * p as the reference forms it (rotation matrix -> Eigen quaternion -> q.rotate(p) + t) agrees with
    R p + t to float rounding for rotations that take each of the four branches of the matrix -> quaternion step."""
    import ctypes
    import os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
    rng = np.random.default_rng(11)

    def rot(axis, ang):
        a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    cases = [rot(rng.normal(size=3), rng.uniform(-1, 1)) for _ in range(50)]           # trace > 0
    for ax in np.eye(3):                                                               # near-half turns: the three
        cases += [rot(ax + 0.05 * rng.normal(size=3), np.pi - e) for e in (0.0, 1e-3, 0.2)]   # diagonal branches
    seen = set()
    for R in cases:
        tr = np.trace(R)
        seen.add("w" if tr > 0 else "xyz"[int(np.argmax(np.diag(R)))])
        Twc = np.concatenate([R, rng.uniform(-3, 3, (3, 1))], axis=1).astype(np.float32)
        for _ in range(20):
            p = rng.uniform(-6, 6, 3).astype(np.float32)
            out = np.zeros(3, np.float32)
            lib.oracle_voxblox_transform(Twc.ctypes.data_as(ctypes.c_void_p), p.ctypes.data_as(ctypes.c_void_p),
                                         out.ctypes.data_as(ctypes.c_void_p))
            want = Twc[:, :3].astype(np.float64) @ p.astype(np.float64) + Twc[:, 3]
            assert np.abs(out - want).max() < 2e-5, (R, p, out, want)
    assert seen == {"w", "x", "y", "z"}


@pytest.mark.gpu
def test_hip_queued_clouds_equal_one_by_one(oracle):
    """plvs_hip_tsdf_voxblox_queue / _flush: InsertCloud may upload and UpdateMap integrate — the layer is that of the
    call-by-call sequence bit for bit; an empty cloud in the queue is a scan too; every reader flushes; clear drops the queue."""
    from plvs_amd.tsdf import TsdfVoxblox
    vs = 0.05
    kfs = make_keyframes(5, seed=61)
    ref = oracle.voxblox(vs)
    hip = TsdfVoxblox(vs, max_blocks=8192)
    for k in kfs[:3]:
        ref.integrate(k["xyz"], rgba_of(k), k["Twc"])
        hip.queue(k["xyz"], rgba_of(k), k["Twc"])
    hip.queue(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[0]["Twc"])
    assert hip.queued() == 4
    hip.flush()
    assert hip.queued() == 0 and hip.last_stats()["points"] == sum(len(k["xyz"]) for k in kfs[:3])
    compare(ref, hip)
    # a reader flushes by itself; a direct integrate takes its place behind the queue
    hip.queue(kfs[3]["xyz"], rgba_of(kfs[3]), kfs[3]["Twc"])
    hip.integrate(kfs[4]["xyz"], rgba_of(kfs[4]), kfs[4]["Twc"])
    ref.integrate(kfs[3]["xyz"], rgba_of(kfs[3]), kfs[3]["Twc"])
    ref.integrate(kfs[4]["xyz"], rgba_of(kfs[4]), kfs[4]["Twc"])
    compare(ref, hip)
    hip.queue(kfs[0]["xyz"], rgba_of(kfs[0]), kfs[0]["Twc"])
    hip.clear()
    assert hip.queued() == 0 and hip.num_chunks() == 0
    hip.close()


@pytest.mark.gpu
def test_mirror_queues_insertions_until_update_map(oracle):
    from plvs_amd.tsdf import PointCloudMapVoxblox
    vs = 0.05
    kfs = make_keyframes(3, seed=62)
    a, b = (PointCloudMapVoxblox(vs, integration_method="simple"),
            PointCloudMapVoxblox(vs, queue_insertions=False, integration_method="simple"))
    for k in kfs:
        a.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
        b.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
    assert a._tsdf.queued() == 3 and b._tsdf.queued() == 0
    ca, cb = a.UpdateMap(), b.UpdateMap()
    assert a._tsdf.queued() == 0 and len(ca) > 1000 and ca.tobytes() == cb.tobytes()
    assert sorted(a.mesh_layer) == sorted(b.mesh_layer)
