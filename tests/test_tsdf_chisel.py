"""Chisel TSDF integrate: oracle self-checks + device arithmetic on the host (CPU),
HIP path vs oracle (GPU).

Bar: the HIP path is BIT-EXACT against the oracle for sdf, weight, kfid and the
u8 colours (the north-star only asks for a float tolerance on sdf/weight; the
device path keeps the reference's per-voxel update order, so no tolerance is
needed: TOL = 0).
"""
import numpy as np
import pytest

from tests.plvs_amd_synth import make_keyframes, TUM1

TOL = 0.0  # absolute tolerance on sdf / weight (bit-exact)


def small_cam(scale=4):
    c = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        c[k] = c[k] / scale
    c["width"] //= scale
    c["height"] //= scale
    return c


def compare_maps(a, b, tol=TOL):
    ia = {tuple(x) for x in a.chunk_ids()}
    ib = {tuple(x) for x in b.chunk_ids()}
    assert ia == ib, f"chunk sets differ: {len(ia)} vs {len(ib)}, only-a {sorted(ia - ib)[:5]}, only-b {sorted(ib - ia)[:5]}"
    for cid in sorted(ia):
        ca, cb = a.get_chunk(*cid), b.get_chunk(*cid)
        if tol == 0.0:
            assert np.array_equal(ca[0].view(np.uint32), cb[0].view(np.uint32)), f"sdf differs in chunk {cid}"
            assert np.array_equal(ca[1].view(np.uint32), cb[1].view(np.uint32)), f"weight differs in chunk {cid}"
        else:
            assert np.allclose(ca[0], cb[0], rtol=0, atol=tol)
            assert np.allclose(ca[1], cb[1], rtol=0, atol=tol)
        assert np.array_equal(ca[2], cb[2]), f"kfid differs in chunk {cid}"
        assert np.array_equal(ca[3], cb[3]), f"colour differs in chunk {cid}"
    return len(ia)


# ------------------------------------------------------------------ CPU
def test_oracle_single_point_known_answer(oracle):
    """One point straight ahead at 0.8 m, identity pose, 5 cm voxels: the ray runs
    along +z through voxel column (0,0,*), straddling the chunk boundary at
    0.8 m; every updated voxel (centre c) holds sdf = |c| * (0.8 / c_z - 1)."""
    m = oracle.chisel(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    m.integrate(np.array([[0.0, 0.0, 0.8]], np.float32), np.array([[255, 128, 0]], np.uint8),
                np.array([7], np.uint32), Twc)
    tr = max(6 * (0.0019 * 0.64 - 0.00152 * 0.8 + 0.001504), 2 * np.sqrt(3) * 0.05)
    assert m.last_visits() in (6, 7, 8)
    assert m.num_chunks() == 2          # z voxels 12..19 straddle chunk z=0 / z=1 (0.8 m)
    seen = 0
    for cid in m.chunk_ids():
        sdf, w, kf, col = m.get_chunk(*cid)
        idx = np.nonzero(w > 0)[0]
        for vid in idx:
            lz = vid >> 8
            cz = (cid[2] * 16 + lz) * 0.05 + 0.025
            cx = 0.025
            expect = np.sqrt(cx * cx * 2 + cz * cz) * (0.8 / cz - 1.0)
            assert abs(sdf[vid] - expect) < 1e-5
            assert abs(sdf[vid]) < tr
            assert abs(w[vid] - 1.0 / (2 * tr)) < 1e-5
            assert kf[vid] == 7
            assert col[vid] == (255 | (128 << 8) | (0 << 16) | (1 << 24))
            seen += 1
    assert seen == m.last_visits()


def test_oracle_skips_near_points_and_empty(oracle):
    m = oracle.chisel(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    m.integrate(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), np.zeros(0, np.uint32), Twc)
    assert m.num_chunks() == 0
    m.integrate(np.array([[0, 0, 0.005], [0, 0, -1.0]], np.float32), np.zeros((2, 3), np.uint8),
                np.zeros(2, np.uint32), Twc)
    assert m.num_chunks() == 0 and m.last_visits() == 0


def test_oracle_colour_weight_saturates(oracle):
    """ColorVoxel weight stops at 254 while the distance weight keeps growing."""
    m = oracle.chisel(0.05)
    Twc = np.eye(4, dtype=np.float32)[:3]
    n = 300
    xyz = np.tile(np.array([[0.01, 0.01, 1.0]], np.float32), (n, 1))
    rgb = np.tile(np.array([[200, 100, 50]], np.uint8), (n, 1))
    m.integrate(xyz, rgb, np.arange(n, dtype=np.uint32), Twc)
    tot = 0
    for cid in m.chunk_ids():
        sdf, w, kf, col = m.get_chunk(*cid)
        for vid in np.nonzero(w > 0)[0]:
            assert (col[vid] >> 24) == 254
            assert kf[vid] == n - 1
            tot += 1
    assert tot * n == m.last_visits()


@pytest.mark.parametrize("res,nkf", [(0.05, 3), (0.02, 2), (0.10, 3)])
def test_device_arithmetic_on_host_matches_oracle(oracle, res, nkf):
    """tsdf_chisel_core.hpp (what the kernels run) compiled with g++ and driven
    sequentially must reproduce the oracle bit for bit."""
    from tests import oracle_lib
    host = oracle_lib._ChiselLike(oracle_lib.load_hostcore(), "hostcore", res)
    ora = oracle.chisel(res)
    for kf in make_keyframes(nkf, cam=small_cam(4), seed=3):
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        host.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        assert ora.last_visits() == host.last_visits() > 0
    assert compare_maps(ora, host) > 3


def test_oracle_shards_partition_the_map(oracle):
    """Owner(chunk) = ChunkHasher(id) mod N: the shards are disjoint and their
    union is the unsharded map, voxel for voxel."""
    kfs = make_keyframes(2, cam=small_cam(4), seed=5)
    full = oracle.chisel(0.05)
    shards = [oracle.chisel(0.05, shard_rank=r, shard_count=3) for r in range(3)]
    for kf in kfs:
        for m in [full] + shards:
            m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    ids = [set(map(tuple, s.chunk_ids())) for s in shards]
    assert not (ids[0] & ids[1]) and not (ids[0] & ids[2]) and not (ids[1] & ids[2])
    assert ids[0] | ids[1] | ids[2] == set(map(tuple, full.chunk_ids()))
    for s in shards:
        for cid in s.chunk_ids():
            a, b = s.get_chunk(*cid), full.get_chunk(*cid)
            assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("res,nkf,scale", [(0.05, 4, 4), (0.02, 2, 4), (0.10, 3, 2), (0.05, 2, 1)])
def test_hip_matches_oracle_bit_exact(oracle, res, nkf, scale):
    from plvs_amd.tsdf import TsdfChisel
    ora = oracle.chisel(res)
    dev = TsdfChisel(res, max_chunks=4096)
    for kf in make_keyframes(nkf, cam=small_cam(scale), seed=11):
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        st = dev.last_stats()
        assert st["visits"] == ora.last_visits()
        assert st["points"] == kf["xyz"].shape[0]
    assert compare_maps(ora, dev) > 3
    dev.close()


@pytest.mark.gpu
def test_hip_batch_equals_sequential_and_oracle(oracle):
    """integrate_batch_dev over K clouds == K single calls == oracle."""
    import torch
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(5, cam=small_cam(2), seed=13)
    ora = oracle.chisel(0.05)
    for kf in kfs:
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    dev = TsdfChisel(0.05, max_chunks=4096)
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    dev.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    torch.cuda.synchronize()
    assert dev.last_stats()["points"] == offsets[-1]
    compare_maps(ora, dev)
    upd = set(map(tuple, dev.updated_chunk_ids()))
    assert upd == set(map(tuple, dev.chunk_ids()))     # first call: every chunk is new and updated
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order_free", [False, True])
def test_hip_queued_clouds_equal_one_by_one(oracle, order_free):
    """plvs_hip_tsdf_chisel_queue + _flush (what the PointCloudMapChisel mirrors do between two UpdateMap calls) against
    one integrate call per key frame: the same map bit for bit in the ordered mode, both within the stated tolerance of
    the oracle in the order-free mode; whatever reads the map flushes the queue."""
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(5, cam=small_cam(2), seed=29)
    ora = oracle.chisel(0.05)
    one = TsdfChisel(0.05, max_chunks=4096, order_free=order_free)
    que = TsdfChisel(0.05, max_chunks=4096, order_free=order_free)
    for kf in kfs[:2]:
        for m in (ora, one):
            m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        que.queue(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    assert que.queued() == 2
    que.flush()
    assert que.queued() == 0 and que.last_stats()["points"] == sum(k["xyz"].shape[0] for k in kfs[:2])
    for kf in kfs[2:]:
        for m in (ora, one):
            m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        que.queue(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    assert que.queued() == 3
    n = que.num_chunks()                 # a reader: flushes
    assert que.queued() == 0 and n == one.num_chunks()
    if order_free:
        for dev in (one, que):
            for cid in map(tuple, ora.chunk_ids()):
                a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
                known = a[1] > 0
                assert np.array_equal(known, b[1] > 0) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
                assert np.abs(a[0][known] - b[0][known]).max(initial=0) <= 2e-5
                assert (np.abs(a[1][known] - b[1][known]) / a[1][known]).max(initial=0) <= 5e-5
    else:
        compare_maps(ora, one)
        compare_maps(one, que)
    que.queue(kfs[0]["xyz"], kfs[0]["rgb"], kfs[0]["kfid"], kfs[0]["Twc"])
    que.clear()                          # the queue goes with the map
    assert que.queued() == 0 and que.num_chunks() == 0
    one.close()
    que.close()


@pytest.mark.gpu
def test_mirror_queues_insertions_until_update_map():
    """PointCloudMapChisel(queue_insertions=True) — InsertCloud uploads, UpdateMap integrates what waits in one batch —
    gives the output cloud of the call-by-call mirror byte for byte (ordered mode)."""
    from plvs_amd.tsdf import PointCloudMapChisel
    kfs = make_keyframes(6, cam=small_cam(2), seed=31)
    outs = []
    for q in (False, True):
        m = PointCloudMapChisel(0.05, max_chunks=4096, queue_insertions=q)
        clouds = []
        for i, kf in enumerate(kfs):
            m.InsertCloud(kf, kf["Twc"])
            if i in (2, 5):
                clouds.append(m.UpdateMap())
        outs.append(clouds)
        m.Clear()
    for a, b in zip(*outs):
        assert len(a) == len(b) > 1000 and a.tobytes() == b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("order_free", [True, False])
def test_hip_far_points_fill_the_tile_tables(oracle, order_free):
    """Key frames of the office stream of bench.py — points up to 5 m away: a tile of 512 rays touches 800 - 2000 voxels —
    at full resolution: tiles that fit the 2048-entry table, tiles that take the 4096-entry pass, tiles the general
    kernel cuts, in a call long enough for the two-pass walk and in single-key-frame calls (the one-pass walk)."""
    import torch
    from tests.synth_scene import make_stream_keyframes
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_stream_keyframes(4, first=1795, threads=4) + make_stream_keyframes(3, first=1000, threads=4)
    ora = oracle.chisel(0.05)
    for kf in kfs:
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    dev = TsdfChisel(0.05, max_chunks=8192, order_free=order_free)
    big = kfs[:5]
    dev.integrate_batch_dev(torch.from_numpy(np.concatenate([k["xyz"] for k in big])).cuda(),
                            torch.from_numpy(np.concatenate([k["rgb"] for k in big])).cuda(),
                            torch.from_numpy(np.concatenate([k["kfid"] for k in big]).astype(np.int32)).cuda(),
                            np.cumsum([0] + [k["xyz"].shape[0] for k in big]).astype(np.int32),
                            torch.from_numpy(np.stack([k["Twc"] for k in big])).cuda())
    for kf in kfs[5:]:
        dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    assert {tuple(c) for c in ora.chunk_ids()} == {tuple(c) for c in dev.chunk_ids()}
    if not order_free:
        compare_maps(ora, dev)
    else:
        for cid in map(tuple, ora.chunk_ids()):
            a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
            known = a[1] > 0
            assert np.array_equal(known, b[1] > 0) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
            assert np.abs(a[0][known] - b[0][known]).max(initial=0) <= 2e-5
            assert (np.abs(a[1][known] - b[1][known]) / a[1][known]).max(initial=0) <= 5e-5
    dev.close()


@pytest.mark.gpu
def test_hip_edge_cases(oracle):
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfChisel
    dev = TsdfChisel(0.05, max_chunks=8)
    Twc = np.eye(4, dtype=np.float32)[:3]
    dev.integrate(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), np.zeros(0, np.uint32), Twc)
    assert dev.num_chunks() == 0
    dev.integrate(np.array([[0, 0, 0.005], [0, 0, -1.0]], np.float32), np.zeros((2, 3), np.uint8),
                  np.zeros(2, np.uint32), Twc)
    assert dev.num_chunks() == 0 and dev.last_stats()["visits"] == 0
    # colour-weight saturation + many updates of the same voxels in one call
    n = 300
    xyz = np.tile(np.array([[0.01, 0.01, 1.0]], np.float32), (n, 1))
    rgb = np.tile(np.array([[200, 100, 50]], np.uint8), (n, 1))
    ora = oracle.chisel(0.05)
    ora.integrate(xyz, rgb, np.arange(n, dtype=np.uint32), Twc)
    dev.integrate(xyz, rgb, np.arange(n, dtype=np.uint32), Twc)
    compare_maps(ora, dev)
    # clear, then pool overflow must fail loudly
    dev.clear()
    assert dev.num_chunks() == 0
    big = make_keyframes(1, seed=1)[0]
    with pytest.raises(_lib.PlvsHipError) as e:
        dev.integrate(big["xyz"], big["rgb"], big["kfid"], big["Twc"])
    assert e.value.code == _lib.PLVS_ERR_CAPACITY
    dev.close()


@pytest.mark.gpu
def test_hip_far_from_the_origin(oracle):
    """Chunk ids come from the integer voxel coordinates; that equals the reference's float lookup while
    |voxel| < 2^20 (proof in tsdf_chisel_core.hpp).  40 km out (800 000 voxels at 5 cm) must still be exact,
    60 km must be refused, not silently different."""
    from plvs_amd import _lib
    from plvs_amd.tsdf import TsdfChisel
    kf = make_keyframes(1, seed=2)[0]
    for shift, ok in ((np.array([40000.0, -39000.0, 38000.0], np.float32), True),
                      (np.array([60000.0, 0.0, 0.0], np.float32), False)):
        Twc = kf["Twc"].copy()
        Twc[:, 3] += shift
        dev = TsdfChisel(0.05)
        if ok:
            ora = oracle.chisel(0.05)
            ora.integrate(kf["xyz"][::7], kf["rgb"][::7], kf["kfid"][::7], Twc)
            dev.integrate(kf["xyz"][::7], kf["rgb"][::7], kf["kfid"][::7], Twc)
            compare_maps(ora, dev)
            assert np.abs(np.asarray(dev.chunk_ids())).max() > 40000
        else:
            with pytest.raises(_lib.PlvsHipError):
                dev.integrate(kf["xyz"][::7], kf["rgb"][::7], kf["kfid"][::7], Twc)
        dev.close()


@pytest.mark.gpu
def test_hip_shards_match_oracle_shards(oracle):
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(2, cam=small_cam(2), seed=17)
    for rank, count in ((0, 2), (1, 2), (2, 3), (5, 8)):     # power-of-two and general owner arithmetic
        ora = oracle.chisel(0.05, shard_rank=rank, shard_count=count)
        dev = TsdfChisel(0.05, max_chunks=4096, shard_rank=rank, shard_count=count)
        for kf in kfs:
            ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
            dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        compare_maps(ora, dev)
        dev.close()


@pytest.mark.gpu
def test_hip_reciprocal_is_correctly_rounded_for_every_significand():
    """chain_runs divides through RN(1/w) computed as v_rcp_f32 + one Newton step; the exact
    quotient recovery needs that reciprocal correctly rounded.  Checked for all 2^23
    significands at the ends and inside the exponent range the kernel admits (2^-20..2^40)."""
    import ctypes
    from plvs_amd import _lib
    f = _lib.lib.plvs_hip_selftest_rcp
    f.argtypes = [ctypes.c_int, ctypes.c_void_p]
    for exponent in (-20, -7, -1, 0, 1, 2, 13, 39):
        bad = ctypes.c_uint32(12345)
        _lib.check(f(exponent, ctypes.byref(bad)))
        assert bad.value == 0, f"exponent {exponent}: {bad.value} significands not correctly rounded"


@pytest.mark.gpu
def test_hip_walk_square_root_and_quotient_are_the_ieee_ones():
    """The single-walk kernel takes sqrt and the quotient of the signed distance without the compiler's range
    scaffolding (tsdf_walk.hpp: sqrt_rn_normal, div_rn_normal); they must agree with sqrtf and `/` bit for bit —
    1.3e9 pseudo-random operand pairs over the exponents the walk meets."""
    import ctypes
    from plvs_amd import _lib
    f = _lib.lib.plvs_hip_selftest_walk_math
    f.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    for seed in (1, 77, 4242, 99991, 123456789):
        bad = (ctypes.c_uint32 * 2)(7, 7)
        _lib.check(f(seed, bad))
        assert bad[0] == 0 and bad[1] == 0, f"seed {seed}: {bad[0]} square roots, {bad[1]} quotients differ"


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [0, 1, 2], ids=["u32_values", "u64_values", "launched_on_a_bound"])
def test_hip_radix_sort_is_a_stable_sort_on_both_scatter_paths(wide):
    """device_utils.hip's sort orders the visit records, the runs and the deform records of both back ends: stable (a
    voxel's items keep their order), pairs intact, on either side of the 2^20-pair switch between the wide-digit
    scatter and the 8-bit LDS-reordering one, for sizes that are not multiples of the tile, few and many distinct
    keys, a bit range that does not start at 0, 32- and 64-bit values; from 2^18 pairs on a pass is one launch with the scan
    chained inside the scatter (radix_onesweep).  wide = 2: the same sort launched on a BOUND of the number of pairs, the
    real number in a device word (radix_sort_pairs_bound: the colour chain of a call that does not wait for its run count)."""
    import ctypes
    from plvs_amd import _lib
    f = _lib.lib.plvs_hip_selftest_radix_sort
    f.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
    cases = [(1, 0, 8), (4097, 0, 12), (100003, 0, 20), ((1 << 20) - 5, 0, 22), (1 << 20, 0, 22), ((1 << 20) + 4097, 0, 26),
             (3333333, 0, 22), (2500000, 0, 5), (1300001, 3, 19), (5000011, 0, 32), (777777, 5, 26),
             (1 << 18, 0, 22), ((1 << 18) - 1, 0, 22), ((1 << 18) + 1, 0, 24), (300000, 2, 18), (20000003, 0, 24)]
    for n, lo, hi in cases:
        if wide == 2 and n > 6000000:      # (the bound doubles the arrays)
            continue
        bad = (ctypes.c_uint32 * 2)(9, 9)
        _lib.check(f(n, lo, hi, wide, 12345 + n, bad))
        assert bad[0] == 0 and bad[1] == 0, f"n={n} bits [{lo},{hi}) wide={wide}: {bad[0]} order, {bad[1]} pair errors"


# Order-free mode (plvs_tsdf_chisel_params.order_free = 1): BASELINE's north star asks for the TSDF
# sdf / weights "within a stated float tolerance" of the reference.  The tolerance, stated here:
ORDER_FREE_SDF_ATOL = 2e-5        # metres (0.04 % of a 5 cm voxel); measured 7e-7 on keyframe streams, 6e-6
                                  # when 6000 identical points pile 6000 sequential roundings into one voxel
ORDER_FREE_WEIGHT_RTOL = 5e-5    # relative
ORDER_FREE_COLOUR_ATOL = 0        # the colour (truncating u8 mean, frozen at weight 254) stays exact


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3])
def test_hip_order_free_mode_is_within_the_stated_tolerance(oracle, batch):
    import torch
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(6, cam=small_cam(2), seed=21)
    ora = oracle.chisel(0.05)
    dev = TsdfChisel(0.05, max_chunks=4096, order_free=True)
    for b0 in range(0, len(kfs), batch):
        part = kfs[b0:b0 + batch]
        for kf in part:
            ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in part])).cuda()
        rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in part])).cuda()
        kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in part]).astype(np.int32)).cuda()
        Twc = torch.from_numpy(np.stack([k["Twc"] for k in part])).cuda()
        offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in part]).astype(np.int32)
        dev.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        torch.cuda.synchronize()
        assert dev.last_stats()["visits"] > 0
    ia = {tuple(x) for x in ora.chunk_ids()}
    assert ia == {tuple(x) for x in dev.chunk_ids()}
    worst = [0.0, 0.0, 0]
    for cid in sorted(ia):
        a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
        known = a[1] > 0
        assert np.array_equal(known, b[1] > 0), "the sets of observed voxels differ"
        assert np.array_equal(a[2], b[2]), "kfid must be exact"
        if not known.any():
            continue
        worst[0] = max(worst[0], float(np.abs(a[0][known] - b[0][known]).max()))
        worst[1] = max(worst[1], float((np.abs(a[1][known] - b[1][known]) / a[1][known]).max()))
        ca, cb = a[3][known], b[3][known]
        assert np.array_equal(np.minimum(ca >> 24, 254), np.minimum(cb >> 24, 254)), "colour weights differ"
        for sh in (0, 8, 16):
            worst[2] = max(worst[2], int(np.abs(((ca >> sh) & 255).astype(int) - ((cb >> sh) & 255).astype(int)).max()))
    print("order-free deviations: sdf %.3g m, weight %.3g rel, colour %d levels" % tuple(worst))
    assert worst[0] <= ORDER_FREE_SDF_ATOL and worst[1] <= ORDER_FREE_WEIGHT_RTOL and worst[2] <= ORDER_FREE_COLOUR_ATOL
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("resolution", [0.05, 0.01])
def test_hip_small_calls_launch_their_colour_chain_on_predicted_sizes(oracle, resolution):
    """A call of up to ~25 key frames launches its colour chain on the run count and chunk count of the call before it
    (integrate_walk_acc: no host read in the middle of the call); when the bounds do not hold the fold skips itself
    and the chain is repeated with the call's own numbers.  Sequences that break the bounds both ways — a few points,
    then whole key frames (the runs exceed the bound; at 1 cm a key frame also adds more chunks than the chunk bound
    allows), then a few points again, then several key frames — must give the oracle's colours exactly."""
    import torch
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(7, cam=small_cam(2), seed=33)
    few = lambda kf, n: dict(kf, xyz=kf["xyz"][:n], rgb=kf["rgb"][:n], kfid=kf["kfid"][:n])
    calls = [[few(kfs[0], 60)], [kfs[0]], [kfs[1]], [few(kfs[2], 30)], [kfs[2], kfs[3], kfs[4]], [kfs[5]], [few(kfs[6], 900)], [kfs[6]]]
    ora = oracle.chisel(resolution)
    dev = TsdfChisel(resolution, max_chunks=16384, order_free=True)
    runs_seen, new_chunks = [], []
    for part in calls:
        for kf in part:
            ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in part])).cuda()
        rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in part])).cuda()
        kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in part]).astype(np.int32)).cuda()
        Twc = torch.from_numpy(np.stack([k["Twc"] for k in part])).cuda()
        offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in part]).astype(np.int32)
        dev.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        runs_seen.append(dev.last_stats()["voxels"])
        new_chunks.append(dev.last_stats()["new_chunks"])
    assert max(runs_seen) > 20 * min(runs_seen), "the calls must differ in size by more than the bound's margin"
    ia = {tuple(x) for x in ora.chunk_ids()}
    assert ia == {tuple(x) for x in dev.chunk_ids()}
    if resolution == 0.01:   # (bound: twice the chunks before the call, or 256 more)
        assert new_chunks[0] < 20 and new_chunks[1] > 2 * new_chunks[0] + 256, "a key frame must add more chunks than the bound allows"
    for cid in sorted(ia):
        a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
        known = a[1] > 0
        assert np.array_equal(known, b[1] > 0)
        assert np.array_equal(a[2], b[2]), "kfid must be exact"
        if known.any():
            assert float(np.abs(a[0][known] - b[0][known]).max()) <= ORDER_FREE_SDF_ATOL
            assert np.array_equal(a[3][known], b[3][known]), f"colours of chunk {cid} differ"
    dev.close()


def _scattered_cloud(n, seed, spread=4.0, zmax=6.0):
    """Points with no spatial coherence at all: consecutive points land in far-apart chunks (a
    tile meets hundreds of chunks: the per-tile chunk cache overflows and falls back)."""
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(0.3, zmax, n)],
                   axis=1).astype(np.float32)
    return xyz, rng.integers(0, 256, (n, 3), dtype=np.uint8), np.full(n, seed, np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("order_free", [False, True])
def test_hip_degenerate_clouds_match_oracle(oracle, order_free):
    """Inputs that stress the tile pipeline: scattered points (chunk-cache overflow), thousands of
    identical points (one voxel collects thousands of visits, groups of hundreds per tile), and a
    cloud whose visit count is a few slots around a multiple of the tile size."""
    from plvs_amd.tsdf import TsdfChisel
    Twc = np.eye(4, dtype=np.float32)[:3]
    clouds = [_scattered_cloud(3000, 1), _scattered_cloud(1500, 2, spread=8.0, zmax=8.0)]
    same = np.tile(np.array([[0.31, -0.22, 1.37]], np.float32), (6000, 1))
    clouds.append((same, np.random.default_rng(3).integers(0, 256, (6000, 3), dtype=np.uint8), np.arange(6000, dtype=np.uint32)))
    ora = oracle.chisel(0.05)
    dev = TsdfChisel(0.05, max_chunks=16384, order_free=order_free)
    for xyz, rgb, kf in clouds:
        ora.integrate(xyz, rgb, kf, Twc)
        dev.integrate(xyz, rgb, kf, Twc)
        assert dev.last_stats()["visits"] == ora.last_visits()
    # visit counts around a tile boundary: trim a cloud until its visit count is 4096 +- 1
    xyz, rgb, kf = _scattered_cloud(2000, 7, spread=3.0, zmax=5.0)
    probe = oracle.chisel(0.05)
    probe.integrate(xyz, rgb, kf, Twc)
    per_point = probe.last_visits() / len(xyz)
    for target in (4095, 4096, 4097):
        lo, hi = 1, len(xyz)
        best = None
        for m in range(max(1, int(target / per_point) - 40), min(len(xyz), int(target / per_point) + 40)):
            p2 = oracle.chisel(0.05)
            p2.integrate(xyz[:m], rgb[:m], kf[:m], Twc)
            if p2.last_visits() >= target:
                best = m
                break
        if best is None:
            continue
        ora.integrate(xyz[:best], rgb[:best], kf[:best], Twc)
        dev.integrate(xyz[:best], rgb[:best], kf[:best], Twc)
    if order_free:
        for cid in {tuple(x) for x in ora.chunk_ids()}:
            a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
            known = a[1] > 0
            assert np.array_equal(known, b[1] > 0) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
            if known.any():
                assert np.abs(a[0][known] - b[0][known]).max() <= ORDER_FREE_SDF_ATOL
                assert (np.abs(a[1][known] - b[1][known]) / a[1][known]).max() <= ORDER_FREE_WEIGHT_RTOL
    else:
        assert compare_maps(ora, dev) > 50
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order_free,shards", [(False, 1), (True, 1), (False, 16)])
def test_hip_long_stretches_without_visits(oracle, order_free, shards):
    """A tile whose visits come from points more than 2^16 apart (skipped points in between: z < 0.01, or —
    on a shard — whole keyframes that only see other ranks' chunks): the per-visit point index no longer
    fits the 16-bit tile-relative form and is recovered from the offsets."""
    from plvs_amd.tsdf import TsdfChisel
    kf = make_keyframes(2, cam=small_cam(2), seed=23)
    a, b = kf[0], kf[1]
    na, nb = 1500, 900
    hole = 70000 if shards == 1 else 3000
    skip = np.tile(np.array([[0.1, 0.1, 0.001]], np.float32), (hole, 1))          # z < 0.01: no ray
    xyz = np.concatenate([a["xyz"][:na], skip, a["xyz"][na:na + nb], skip, a["xyz"][na + nb:na + 2 * nb]])
    rgb = np.concatenate([a["rgb"][:na], np.zeros((hole, 3), np.uint8), a["rgb"][na:na + nb],
                          np.zeros((hole, 3), np.uint8), a["rgb"][na + nb:na + 2 * nb]])
    kfid = np.arange(xyz.shape[0], dtype=np.uint32)                                 # kfid = point index: exposes a wrong point
    for rank in ((0,) if shards == 1 else (0, 15)):     # rank 15 of 16 owns ~500 visits per copy: tiles ~70 000 points wide
        ora = oracle.chisel(0.05, shard_rank=rank, shard_count=shards)
        dev = TsdfChisel(0.05, max_chunks=4096, shard_rank=rank, shard_count=shards, order_free=order_free)
        if shards > 1:
            # many repetitions of the cloud: a rank's 4096-visit tiles stretch over several copies
            reps = 40
            xyz_r, rgb_r = np.tile(xyz, (reps, 1)), np.tile(rgb, (reps, 1))
            kfid_r = np.arange(xyz_r.shape[0], dtype=np.uint32)
            ora.integrate(xyz_r, rgb_r, kfid_r, a["Twc"])
            dev.integrate(xyz_r, rgb_r, kfid_r, a["Twc"])
        else:
            ora.integrate(xyz, rgb, kfid, a["Twc"])
            dev.integrate(xyz, rgb, kfid, a["Twc"])
        ora.integrate(b["xyz"], b["rgb"], b["kfid"], b["Twc"])
        dev.integrate(b["xyz"], b["rgb"], b["kfid"], b["Twc"])
        if order_free:
            for cid in {tuple(x) for x in ora.chunk_ids()}:
                p, q = ora.get_chunk(*cid), dev.get_chunk(*cid)
                known = p[1] > 0
                assert np.array_equal(known, q[1] > 0) and np.array_equal(p[2], q[2]) and np.array_equal(p[3], q[3])
                if known.any():
                    assert np.abs(p[0][known] - q[0][known]).max() <= ORDER_FREE_SDF_ATOL
        else:
            assert compare_maps(ora, dev) >= 1      # a 1/16 shard of this small scene is one or two chunks
        dev.close()


# ------------------------------------------------------------------ carving (T7)
def _depth_image(cam, kind, seed=0):
    h, w = cam["height"], cam["width"]
    rng = np.random.default_rng(seed)
    if kind == "far":          # everything measured far away: whatever was mapped in front is carved
        d = np.full((h, w), 4.2, np.float32)
    elif kind == "mixed":      # some pixels far, some close (nothing to carve there), some without a measurement
        d = rng.uniform(0.4, 4.8, (h, w)).astype(np.float32)
        d[rng.random((h, w)) < 0.15] = np.nan
    else:                      # "near": the surface moved closer: nothing in front of it
        d = np.full((h, w), 0.3, np.float32)
    return d


def test_oracle_carving_properties(oracle):
    cam = small_cam(4)
    kfs = make_keyframes(4, cam=cam, seed=31)
    m = oracle.chisel(0.05)
    for kf in kfs:
        m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    before = {tuple(c): m.get_chunk(*c) for c in m.chunk_ids()}
    known_before = sum(int((v[1] > 0).sum()) for v in before.values())
    # a surface right in front of the camera: no voxel is more than truncation + 0.05 in front of it
    n_near, _ = m.carve(_depth_image(cam, "near"), cam["fx"], cam["fy"], cam["cx"], cam["cy"], kfs[0]["Twc"])
    assert n_near == 0
    n_far, ids = m.carve(_depth_image(cam, "far"), cam["fx"], cam["fy"], cam["cx"], cam["cy"], kfs[0]["Twc"])
    assert n_far == len(ids) > 0
    after = {tuple(c): m.get_chunk(*c) for c in m.chunk_ids()}
    assert set(after) == set(before)                 # carving never creates or drops chunks
    known_after = sum(int((v[1] > 0).sum()) for v in after.values())
    assert known_after < known_before
    touched = {tuple(i) for i in ids}
    for cid, (sdf, w, kf, col) in after.items():
        b = before[cid]
        changed = (w != b[1])
        assert changed.any() == (cid in touched)
        # a carved voxel is reset: sdf 99999, weight 0, kfid 0, colour untouched; only voxels with sdf < 1e-5 go
        assert np.all(sdf[changed] == 99999.0) and np.all(w[changed] == 0) and np.all(kf[changed] == 0)
        assert np.all(b[0][changed] < 1e-5) and np.array_equal(col, b[3])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["far", "mixed", "near"])
def test_hip_carving_matches_oracle(oracle, kind):
    from plvs_amd.tsdf import TsdfChisel
    cam = small_cam(2)
    kfs = make_keyframes(5, cam=cam, seed=33)
    ora = oracle.chisel(0.05)
    dev = TsdfChisel(0.05, max_chunks=4096)
    for kf in kfs:
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    for step, pose in enumerate((kfs[1]["Twc"], kfs[4]["Twc"])):
        depth = _depth_image(cam, kind, seed=step)
        want_n, want_ids = ora.carve(depth, cam["fx"], cam["fy"], cam["cx"], cam["cy"], pose)
        got_n = dev.carve(depth, cam["fx"], cam["fy"], cam["cx"], cam["cy"], pose)
        assert got_n == want_n
        assert {tuple(i) for i in dev.updated_chunk_ids()} == {tuple(i) for i in want_ids}
        compare_maps(ora, dev)
    # integrating on top of a carved map still agrees
    ora.integrate(kfs[2]["xyz"], kfs[2]["rgb"], kfs[2]["kfid"], kfs[2]["Twc"])
    dev.integrate(kfs[2]["xyz"], kfs[2]["rgb"], kfs[2]["kfid"], kfs[2]["Twc"])
    compare_maps(ora, dev)
    dev.close()


def test_shard_of_is_hash_mod_count():
    """shard_of (32-bit arithmetic, mask for powers of two) == ChunkHasher(id) % N in size_t arithmetic."""
    import ctypes
    from tests import oracle_lib
    lib = oracle_lib.load_hostcore()
    lib.hostcore_shard_of.argtypes = [ctypes.c_ulonglong, ctypes.c_int]
    rng = np.random.default_rng(11)
    hs = [0, 1, 2**32 - 1, 2**32, 2**64 - 1, 2**63, 73856093 * 5 ^ 19349663 * 7] + \
        [int(x) for x in rng.integers(0, 2**64, 4000, dtype=np.uint64)]
    for count in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 31, 64, 1000, 65521):
        for h in hs:
            assert lib.hostcore_shard_of(h, count) == h % count, (h, count)


@pytest.mark.parametrize("count", [3, 4, 8, 16])
def test_shard_cull_never_drops_an_owned_visit(count):
    """walk_may_touch_owned (the cull ahead of the walk on sharded maps) is conservative: no ray with a
    visit this rank owns is culled — and it does cull a useful share of the others."""
    import ctypes
    from tests import oracle_lib
    lib = oracle_lib.load_hostcore()
    lib.hostcore_cull_violations.restype = ctypes.c_longlong
    kfs = make_keyframes(3, cam=small_cam(4), seed=9)
    # far from the origin too: negative coordinates, large hashes
    far = np.array([-1234.56, 789.01, -33.3], np.float32)
    culled_total = owned_total = 0
    for rank in range(count):
        host = oracle_lib._ChiselLike(lib, "hostcore", 0.05, shard_rank=rank, shard_count=count)
        for kf in kfs:
            for shift in (None, far):
                Twc = kf["Twc"].copy()
                if shift is not None:
                    Twc[:, 3] += shift
                culled = ctypes.c_longlong(0)
                owned = ctypes.c_longlong(0)
                xyz = np.ascontiguousarray(kf["xyz"], np.float32)
                bad = lib.hostcore_cull_violations(host.h, xyz.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.c_int(xyz.shape[0]),
                                                   np.ascontiguousarray(Twc, np.float32).ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.byref(culled), ctypes.byref(owned))
                assert bad == 0
                culled_total += culled.value
                owned_total += owned.value
    assert owned_total > 0
    if count >= 8:
        assert culled_total > 0


def test_integer_chunk_ids_equal_the_float_lookup():
    """v >> 4 == floor((v * res + res / 2) * (1 / (16 res))) for every |v| the kernels accept
    (kVoxelCoordLimit): dense near the chunk boundaries, where a rounding slip would show."""
    import ctypes
    from tests import oracle_lib
    lib = oracle_lib.load_hostcore()
    lib.hostcore_chunk_id_mismatches.restype = ctypes.c_longlong
    lim = 1048576 - 128
    rng = np.random.default_rng(4)
    edges = (rng.integers(-lim // 16, lim // 16, 200000)[:, None] * 16 + np.array([-1, 0, 1, 15, 16])[None, :]).ravel()
    v = np.concatenate([np.arange(-70000, 70000), edges, rng.integers(-lim + 1, lim, 500000),
                        np.arange(lim - 5000, lim), np.arange(-lim + 1, -lim + 5000)]).astype(np.int32)
    v = np.ascontiguousarray(v[np.abs(v) < lim])
    for res in (0.05, 0.02, 0.10, 0.04, 0.0123, 0.25):
        bad = lib.hostcore_chunk_id_mismatches(ctypes.c_float(res), v.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(v.size))
        assert bad == 0, res
