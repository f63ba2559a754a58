"""Ray-sharded multi-GPU integrate of the chisel back end (plvs_hip_tsdf_chisel_shard_walk / _pack / _apply),
run as VIRTUAL ranks on one device: N handles, the all-to-all emulated with tensor slices.  The partial sums are
integers, so the union of the shards must equal the single-device order-free map bit for bit — sdf, weight, kfid
and colour — for any N."""
import numpy as np
import pytest
import torch

from plvs_amd.shard import owner_of
from tests.synth_scene import make_keyframes


def _batch(kfs):
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    return xyz, rgb, kfid, offsets, Twc


WIDTHS = (8, 8, 6)   # int32 words of a descriptor, a voxel sum, a colour-run record


def virtual_all_to_all(counts, bufs):
    """counts[src][dst] = (segments, records, runs); bufs[src] = the three send buffers grouped by destination
    -> per destination the three receive buffers grouped by source, and its receive counts."""
    world = len(counts)
    out = []
    for dst in range(world):
        parts, rc = [[], [], []], np.zeros((world, 3), np.int64)
        for src in range(world):
            off = counts[src][:dst].sum(axis=0)
            rc[src] = counts[src][dst]
            for k in range(3):
                parts[k].append(bufs[src][k][off[k]:off[k] + rc[src][k]])
        out.append(tuple(torch.cat(p).contiguous() for p in parts) + (rc,))
    return out


def send_buffers(t, c):
    bufs = tuple(torch.zeros((int(c[:, k].sum()), w), dtype=torch.int32, device="cuda") for k, w in enumerate(WIDTHS))
    t.shard_pack(*bufs)
    return bufs


def sharded_step(ranks, xyz, rgb, kfid, offsets, Twc):
    counts = [t.shard_walk(xyz, offsets, Twc) for t in ranks]
    bufs = [send_buffers(t, c) for t, c in zip(ranks, counts)]
    torch.cuda.synchronize()
    for t, (seg, rec, run, rc) in zip(ranks, virtual_all_to_all(counts, bufs)):
        t.shard_apply(seg, rec, run, rc, rgb, kfid)
    sat = [t.shard_saturated() for t in ranks]   # the all-gather of the newly saturated voxels
    for t in ranks:
        for lst in sat:
            if lst.shape[0]:
                t.shard_note_saturated(lst)
    torch.cuda.synchronize()
    return counts


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_hip_ray_sharded_integrate_equals_the_single_device_map(world):
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(12, max_depth=5.0, seed=3)
    single = TsdfChisel(0.05, max_chunks=4096, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=4096, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    walked = 0
    runs_sent = []
    for b0 in range(0, len(kfs), 5):   # batches of 5, 5 and 2 keyframes: later calls meet half-saturated colours
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b0 + 5])
        single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        want_visits = single.last_stats()["visits"]
        counts = sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
        runs_sent.append(int(sum(c[:, 2].sum() for c in counts)))
        assert sum(t.last_stats()["visits"] for t in ranks) == want_visits, "every visit is walked by exactly one rank"
        walked += want_visits
    assert walked > 0
    ids = {tuple(x) for x in single.chunk_ids()}
    seen = {}
    for r, t in enumerate(ranks):
        for cid in (tuple(x) for x in t.chunk_ids()):
            assert cid not in seen, "a chunk lives on one rank"
            assert owner_of(np.array([cid]), world)[0] == r, "on the rank the three-prime hash names"
            seen[cid] = r
    assert set(seen) == ids
    for cid in sorted(ids):
        a, b = single.get_chunk(*cid), ranks[seen[cid]].get_chunk(*cid)
        for name, x, y in zip(("sdf", "weight", "kfid", "colour"), a, b):
            assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y), f"{name} of chunk {cid} differs"
    saturated = sum(int(((single.get_chunk(*cid)[3] >> 24) >= 254).sum()) for cid in ids)
    assert saturated > 0, "the scene must drive some colours to saturation for the test to cover the feedback"
    assert runs_sent[1] < runs_sent[0], "walkers stop sending the runs of voxels reported saturated"
    for t in ranks + [single]:
        t.close()


def sharded_step_messages(ranks, xyz, rgb, kfid, offsets, Twc, rows):
    """sharded_step with the feedback in its message form (what plvs_amd.shard.sharded_integrate does): fixed-size
    messages of `rows` voxels + a length row, gathered, noted without a host read.  Returns the voxels announced."""
    world = len(ranks)
    counts = [t.shard_walk(xyz, offsets, Twc) for t in ranks]
    bufs = [send_buffers(t, c) for t, c in zip(ranks, counts)]
    torch.cuda.synchronize()
    for t, (seg, rec, run, rc) in zip(ranks, virtual_all_to_all(counts, bufs)):
        t.shard_apply(seg, rec, run, rc, rgb, kfid)
    gathered = torch.full((world * (rows + 1), 4), -7, dtype=torch.int32, device="cuda")
    for r, t in enumerate(ranks):
        t.shard_saturated_message(gathered[r * (rows + 1):(r + 1) * (rows + 1)], rows)
    for t in ranks:
        t.shard_note_gathered(gathered, world, rows)
    torch.cuda.synchronize()
    g = gathered.cpu().numpy().reshape(world, rows + 1, 4)
    assert all(0 <= g[r, rows, 0] <= rows and not g[r, rows, 1:].any() for r in range(world))
    return [tuple(v) for r in range(world) for v in g[r, : g[r, rows, 0]].tolist()]


@pytest.mark.gpu
def test_hip_saturation_feedback_as_fixed_size_messages_with_a_waiting_list():
    """The feedback without host reads: with messages of 40 voxels most of a call's saturated voxels wait in the handle
    and go out over the following steps — every voxel is announced exactly once, the same set the list form
    (shard_saturated) reports, and the maps stay bit-identical to the single-device one."""
    from plvs_amd.tsdf import TsdfChisel
    world, rows = 3, 40
    kfs = make_keyframes(12, max_depth=5.0, seed=3)
    single = TsdfChisel(0.05, max_chunks=4096, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=4096, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    lists = [TsdfChisel(0.05, max_chunks=4096, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    announced, listed = [], []
    for b0 in range(0, len(kfs), 4):
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b0 + 4])
        single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        announced += sharded_step_messages(ranks, xyz, rgb, kfid, offsets, Twc, rows)
        counts = [t.shard_walk(xyz, offsets, Twc) for t in lists]
        bufs = [send_buffers(t, c) for t, c in zip(lists, counts)]
        for t, (seg, rec, run, rc) in zip(lists, virtual_all_to_all(counts, bufs)):
            t.shard_apply(seg, rec, run, rc, rgb, kfid)
        for t in lists:
            listed += [tuple(v) for v in t.shard_saturated().cpu().numpy().tolist()]
    assert len(listed) > 4 * world * rows, "the scene must saturate more voxels than the messages carry"
    assert len(announced) < len(listed), "so some are still waiting"
    e = np.zeros(1, np.int32)
    for _ in range(len(listed) // rows + 2):   # empty steps drain the waiting lists
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[:1])
        announced += sharded_step_messages(ranks, xyz[:0], rgb[:0], kfid[:0], e, Twc[:0], rows)
    assert len(announced) == len(set(announced)), "a voxel is announced once"
    assert set(announced) == set(listed)
    got = {}
    for t in ranks:
        for cid in (tuple(x) for x in t.chunk_ids()):
            got[cid] = t.get_chunk(*cid)
    assert set(got) == {tuple(x) for x in single.chunk_ids()}
    for cid, b in got.items():
        a = single.get_chunk(*cid)
        assert all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b))
    for t in ranks + lists + [single]:
        t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_hip_ray_sharded_colour_runs_with_many_and_with_long_spans(world):
    """The wire form of a colour run is a list of ray spans, six per record, 128 rays per span at most: clouds whose
    neighbouring points alternate between two distant surfaces (a voxel's rays are every second point of a tile: dozens
    of one-ray spans, several records per run) and a point repeated 700 times (every voxel on its ray is seen by all 512
    rays of a tile: one stretch cut into four spans) must still give the single-device colours bit for bit."""
    from plvs_amd.tsdf import TsdfChisel
    kf = make_keyframes(1, max_depth=5.0, seed=11)[0]
    n = kf["xyz"].shape[0]
    a, b = np.arange(0, 1500), np.arange(n // 2, n // 2 + 1500)
    mixed = np.empty(3000, np.int64)
    mixed[0::2], mixed[1::2] = a, b
    idx = np.concatenate([mixed, np.full(700, n // 3), np.arange(2000, 2600)])
    rng = np.random.default_rng(5)
    sub = dict(kf, xyz=kf["xyz"][idx], rgb=rng.integers(0, 256, (idx.shape[0], 3)).astype(np.uint8), kfid=kf["kfid"][idx])
    single = TsdfChisel(0.05, max_chunks=1024, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=1024, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    xyz, rgb, kfid, offsets, Twc = _batch([sub])
    records = 0
    for _ in range(2):   # (the second call folds onto half-filled colour weights)
        single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        counts = sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
        records += int(sum(c[:, 2].sum() for c in counts))
    got = {}
    for t in ranks:
        for cid in (tuple(x) for x in t.chunk_ids()):
            got[cid] = t.get_chunk(*cid)
    assert set(got) == {tuple(x) for x in single.chunk_ids()}
    for cid, y in got.items():
        x = single.get_chunk(*cid)
        assert all(np.array_equal(p.view(np.uint32) if p.dtype == np.float32 else p,
                                  q.view(np.uint32) if q.dtype == np.float32 else q) for p, q in zip(x, y)), cid
    voxels = sum(int((single.get_chunk(*cid)[1] > 0).sum()) for cid in got)
    assert records > 3 * voxels, "the alternating points must have produced runs of several records"
    for t in ranks + [single]:
        t.close()


@pytest.mark.gpu
def test_hip_ray_sharded_integrate_with_more_ranks_than_tiles_and_empty_calls():
    from plvs_amd.tsdf import TsdfChisel
    kf = make_keyframes(1, max_depth=5.0, seed=5)[0]
    sub = dict(kf, xyz=kf["xyz"][:700], rgb=kf["rgb"][:700], kfid=kf["kfid"][:700])   # two tiles, four ranks
    world = 4
    single = TsdfChisel(0.05, max_chunks=1024, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=1024, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    xyz, rgb, kfid, offsets, Twc = _batch([sub])
    single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    counts = sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
    assert counts[2].sum() == 0 and counts[3].sum() == 0, "ranks without a tile send nothing"
    got = {}
    for t in ranks:
        for cid in (tuple(x) for x in t.chunk_ids()):
            got[cid] = t.get_chunk(*cid)
    assert set(got) == {tuple(x) for x in single.chunk_ids()}
    for cid, b in got.items():
        a = single.get_chunk(*cid)
        assert all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b))
    # an empty call is a no-op on every rank
    e = np.zeros(1, np.int32)
    for t in ranks:
        c = t.shard_walk(xyz[:0], e, Twc[:0])
        assert c.sum() == 0
        empty = tuple(torch.zeros((0, w), dtype=torch.int32, device="cuda") for w in WIDTHS)
        t.shard_pack(*empty)
        t.shard_apply(*empty, np.zeros((world, 3), np.int64), rgb[:0], kfid[:0])
    for t in ranks + [single]:
        t.close()


@pytest.mark.gpu
def test_hip_ray_sharded_integrate_through_torch_distributed_and_rccl_at_world_one():
    """The two real transports with one rank (a rank sends to itself): plvs_amd.shard.sharded_integrate over
    torch.distributed ("nccl" = RCCL) and plvs_hip_tsdf_chisel_integrate_sharded over an ncclComm_t of its own."""
    import ctypes
    import os
    import socket

    import torch.distributed as dist

    from plvs_amd import _lib
    from plvs_amd.shard import sharded_integrate
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(6, max_depth=5.0, seed=7)
    single = TsdfChisel(0.05, max_chunks=2048, order_free=True)
    via_torch = TsdfChisel(0.05, max_chunks=2048, order_free=True)
    via_rccl = TsdfChisel(0.05, max_chunks=2048, order_free=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    rccl = ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)
    uid = (ctypes.c_char * 128)()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()

    class _Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclCommInitRank.argtypes = [ctypes.c_void_p, ctypes.c_int, _Uid, ctypes.c_int]
    u = _Uid()
    ctypes.memmove(ctypes.byref(u), uid, 128)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, u, 0) == 0
    f = _lib.lib.plvs_hip_tsdf_chisel_integrate_sharded
    f.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 2
    try:
        for b0 in range(0, len(kfs), 3):
            xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b0 + 3])
            single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
            sharded_integrate(via_torch, xyz, rgb, kfid, offsets, Twc)
            _lib.check(f(via_rccl._h, comm, _lib.t_ptr(xyz), _lib.t_ptr(rgb), _lib.t_ptr(kfid), _lib.np_ptr(offsets),
                         offsets.shape[0] - 1, _lib.t_ptr(Twc), _lib.current_stream_ptr()))
            torch.cuda.synchronize()
            assert via_torch.last_stats()["visits"] == single.last_stats()["visits"] == via_rccl.last_stats()["visits"]
        ids = sorted(tuple(x) for x in single.chunk_ids())
        for other in (via_torch, via_rccl):
            assert sorted(tuple(x) for x in other.chunk_ids()) == ids
            for cid in ids:
                for x, y in zip(single.get_chunk(*cid), other.get_chunk(*cid)):
                    assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                          y.view(np.uint32) if y.dtype == np.float32 else y)
        # meshing through the same two transports (one rank: nothing is foreign, the collectives still run)
        from plvs_amd.shard import sharded_mesh_chunks
        todo = np.array(ids, np.int32)
        want = single.mesh_chunks(todo)
        got, fetched = sharded_mesh_chunks(via_torch, todo)
        assert fetched == 0 and via_rccl.halo_gather(comm, todo) == 0
        got2 = via_rccl.mesh_chunks(todo)
        for name in ("vertices", "normals", "colors", "kfids", "chunk_first"):
            assert want[name].tobytes() == got[name].tobytes() == got2[name].tobytes(), name
        assert len(want["vertices"]) > 5000
        # and a voxblox map through sharded_mesh_blocks (one rank: no neighbour is foreign, the collectives still run)
        from plvs_amd.shard import sharded_mesh_blocks
        from plvs_amd.tsdf import TsdfVoxblox
        vb = TsdfVoxblox(0.05)
        k0 = kfs[0]
        vb.integrate(k0["xyz"], np.concatenate([k0["rgb"], np.full((len(k0["rgb"]), 1), 255, np.uint8)], 1), k0["Twc"])
        bids = np.ascontiguousarray(vb.chunk_ids(), np.int32)
        a, fetched = sharded_mesh_blocks(vb, bids)
        b = vb.mesh_blocks(bids)
        assert fetched == 0 and all(a[n].tobytes() == b[n].tobytes() for n in ("vertices", "normals", "colors", "block_first"))
        assert vb.halo_gather(comm, bids) == 0          # the same exchange behind the C ABI over the ncclComm_t
        vb.close()
        # the ray-sharded voxblox integrate through the same two transports
        from plvs_amd.shard import sharded_integrate_voxblox
        vs, vt, vr = TsdfVoxblox(0.05), TsdfVoxblox(0.05), TsdfVoxblox(0.05)
        fv = _lib.lib.plvs_hip_tsdf_voxblox_integrate_sharded
        fv.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 2
        for b0 in range(0, len(kfs), 3):
            part = kfs[b0:b0 + 3]
            xyz, rgb, kfid, offsets, Twc = _batch(part)
            rgba = torch.cat([rgb, torch.full((rgb.shape[0], 1), 255, dtype=torch.uint8, device="cuda")], dim=1).contiguous()
            vs.integrate_batch_dev(xyz, rgba, offsets, Twc)
            sharded_integrate_voxblox(vt, xyz, rgba, offsets, Twc)
            _lib.check(fv(vr._h, comm, _lib.t_ptr(xyz), _lib.t_ptr(rgba), _lib.np_ptr(offsets), offsets.shape[0] - 1,
                          _lib.t_ptr(Twc), _lib.current_stream_ptr()))
            torch.cuda.synchronize()
            assert vs.last_stats()["visits"] == vt.last_stats()["visits"] == vr.last_stats()["visits"] > 0
        bids = sorted(tuple(x) for x in vs.chunk_ids())
        for other in (vt, vr):
            assert sorted(tuple(x) for x in other.chunk_ids()) == bids
            for bid in bids:
                for x, y in zip(vs.get_chunk(*bid), other.get_chunk(*bid)):
                    assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                          y.view(np.uint32) if y.dtype == np.float32 else y)
        for t in (vs, vt, vr):
            t.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
        dist.destroy_process_group()
        for t in (single, via_torch, via_rccl):
            t.close()


@pytest.mark.gpu
def test_hip_order_free_apply_in_parts_gives_the_same_map():
    """A busy chunk is applied in parts (sums through global accumulators, the last part applies): forced here with
    tiny thresholds, on one device and on the owners of a ray-sharded map; every variant must be bit-identical."""
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(6, max_depth=5.0, seed=9)
    plain = TsdfChisel(0.05, max_chunks=2048, order_free=True)
    parts = TsdfChisel(0.05, max_chunks=2048, order_free=True)
    parts.set_apply_parts(8, 16)          # nearly every chunk has more than 16 segments: few accumulator sets at first
    world = 3
    ranks = [TsdfChisel(0.05, max_chunks=2048, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    for t in ranks:
        t.set_apply_parts(4, 8)
    for b0 in range(0, len(kfs), 2):
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b0 + 2])
        plain.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        parts.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
        assert parts.last_stats() == plain.last_stats()
    ids = sorted(tuple(x) for x in plain.chunk_ids())
    assert sorted(tuple(x) for x in parts.chunk_ids()) == ids
    owner = {}
    for r, t in enumerate(ranks):
        for cid in (tuple(x) for x in t.chunk_ids()):
            owner[cid] = t
    assert sorted(owner) == ids
    for cid in ids:
        a = plain.get_chunk(*cid)
        for other in (parts.get_chunk(*cid), owner[cid].get_chunk(*cid)):
            for x, y in zip(a, other):
                assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                      y.view(np.uint32) if y.dtype == np.float32 else y), cid
    for t in ranks + [plain, parts]:
        t.close()


def _nbhd27(ids):
    s = set()
    for c in ids:
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    s.add((int(c[0]) + dx, int(c[1]) + dy, int(c[2]) + dz))
    return sorted(s)


def virtual_halo_round(ranks, missing):
    """The meshing halo between virtual ranks: every rank's missing ids go to their owners' halo_export, the answers
    into the asker's halo_import (what plvs_amd.shard.halo_round does over a process group).  -> chunks moved."""
    world, moved = len(ranks), 0
    for r, t in enumerate(ranks):
        if not len(missing[r]):
            continue
        own = owner_of(missing[r], world)
        assert not (own == r).any(), "a rank never misses a chunk of its own"
        for q in range(world):
            ids = np.ascontiguousarray(missing[r][own == q], np.int32)
            if not len(ids):
                continue
            d_ids = torch.from_numpy(ids).cuda()
            found = torch.zeros(len(ids), dtype=torch.int32, device="cuda")
            ranks[q].halo_lookup(d_ids, found)
            nfound = int(found.sum().item())
            payload = torch.empty((nfound, t.HALO_WORDS), dtype=torch.int32, device="cuda")
            ranks[q].halo_export(d_ids, found, payload)
            t.halo_import(d_ids, found, payload)
            moved += nfound
    torch.cuda.synchronize()
    return moved


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5])
def test_hip_sharded_meshes_equal_the_single_device_meshes(world):
    """Meshing a sharded map: each rank meshes ITS chunks of meshesToUpdate, fetching the neighbour chunks the cubes,
    the colour interpolation and the gradient normals read from their owners, in rounds, until nobody misses
    anything.  Every chunk's mesh must be the single-device one byte for byte; the halo goes away with the next
    integrate call and leaves the pool as it was."""
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(7, max_depth=5.0, seed=3)
    single = TsdfChisel(0.05, max_chunks=4096, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=4096, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    for phase, (b0, b1) in enumerate(((0, 5), (5, 7))):
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b1])
        single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
        updated = [tuple(int(v) for v in c) for t in ranks for c in t.updated_chunk_ids()]
        assert sorted(updated) == sorted(tuple(int(v) for v in c) for c in single.updated_chunk_ids())
        todo = np.array(_nbhd27(updated), np.int32)                       # Chisel::meshesToUpdate
        own = owner_of(todo, world)
        mine = [np.ascontiguousarray(todo[own == r]) for r in range(world)]
        before = [t.num_chunks() for t in ranks]
        rounds, moved = 0, 0
        assert ranks[0].mesh_chunks(mine[0], halo_ok=True) is None, "the first pass reaches for other ranks' chunks"
        while True:
            missing = [t.halo_missing() if t.mesh_probe(mine[r]) else np.zeros((0, 3), np.int32) for r, t in enumerate(ranks)]
            if not any(len(m) for m in missing):
                break
            rounds += 1
            assert rounds <= 4, "the halo settles in a few rounds"
            moved += virtual_halo_round(ranks, missing)
        assert rounds >= 1 and moved > 0
        result = [t.mesh_chunks(mine[r]) for r, t in enumerate(ranks)]
        want = single.mesh_chunks(todo)
        wf = want["chunk_first"]
        pos = {tuple(int(v) for v in c): i for i, c in enumerate(todo)}
        total = 0
        for r in range(world):
            gf = result[r]["chunk_first"]
            for j, c in enumerate(mine[r]):
                i = pos[tuple(int(v) for v in c)]
                a, b, a2, b2 = int(wf[i]), int(wf[i + 1]), int(gf[j]), int(gf[j + 1])
                assert b - a == b2 - a2, (c, b - a, b2 - a2)
                for name in ("vertices", "normals", "colors", "kfids"):
                    assert want[name][a:b].tobytes() == result[r][name][a2:b2].tobytes(), (name, c)
                total += b - a
        assert total == len(want["vertices"]) > 8000
        # a rank asked for another rank's chunk owns nothing for it
        foreign = np.ascontiguousarray(todo[own == 1][:3])
        m0 = ranks[0].mesh_chunks(foreign, halo_ok=True)
        assert m0 is not None and len(m0["vertices"]) == 0
        assert [t.num_chunks() for t in ranks] == before, "ghosts are not chunks of the map"
        if phase == 0:
            continue            # the next integrate call drops the halo and re-uses its pool slots
        for t in ranks:
            t.halo_clear()
    # after the halo is gone the shards still hold exactly the single-device map
    ids = {tuple(x) for x in single.chunk_ids()}
    for t in ranks:
        for cid in (tuple(x) for x in t.chunk_ids()):
            a, b = single.get_chunk(*cid), t.get_chunk(*cid)
            assert all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                      y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b)), cid
            ids.discard(cid)
    assert not ids
    for t in ranks + [single]:
        t.close()
