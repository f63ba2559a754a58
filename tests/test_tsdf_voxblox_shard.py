"""Ray-sharded multi-GPU integrate of the voxblox back end (plvs_hip_tsdf_voxblox_shard_walk / _pack / _apply; "simple"),
run as VIRTUAL ranks on one device: N handles, the all-to-all emulated with tensor slices.  updateTsdfVoxel is order
dependent, and the owners apply the visits in the reference's order: every rank's shard must equal the oracle's map of
that shard (the sequential loop with the owner filter) bit for bit — distance, weight, colour — for any N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.plvs_amd_synth import make_keyframes
from tests.test_tsdf_voxblox import compare, rgba_of, small_cam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(kfs):
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgba = torch.from_numpy(np.concatenate([rgba_of(k) for k in kfs])).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    return xyz, rgba, offsets, Twc


def sharded_step(ranks, xyz, rgba, offsets, Twc):
    """walk + pack on every virtual rank, the all-to-all by tensor slices, apply.  Returns the send counts."""
    world = len(ranks)
    counts = [t.shard_walk(xyz, offsets, Twc) for t in ranks]
    sends = []
    for t, c in zip(ranks, counts):
        buf = torch.zeros((int(c.sum()), 4), dtype=torch.int32, device="cuda")
        t.shard_pack(buf)
        sends.append(buf)
    torch.cuda.synchronize()
    for dst, t in enumerate(ranks):
        parts, rc = [], np.zeros(world, np.int64)
        for src in range(world):
            off = int(counts[src][:dst].sum())
            rc[src] = counts[src][dst]
            parts.append(sends[src][off:off + int(rc[src])])
        t.shard_apply(torch.cat(parts).contiguous(), rc, xyz, rgba, offsets, Twc)
    torch.cuda.synchronize()
    return counts


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("carving", [False, True])
def test_hip_ray_sharded_voxblox_integrate_equals_the_oracle_shard_by_shard(oracle, world, carving):
    from plvs_amd.tsdf import TsdfVoxblox
    kfs = make_keyframes(7, cam=small_cam(2), seed=29, max_depth=7.0)   # (depths beyond the 5 m ray limit)
    ranks = [TsdfVoxblox(0.05, use_carving=carving, max_blocks=8192, shard_rank=r, shard_count=world) for r in range(world)]
    oras = [oracle.voxblox(0.05, carving=carving, shard_rank=r, shard_count=world) for r in range(world)]
    single = TsdfVoxblox(0.05, use_carving=carving, max_blocks=8192)
    sent = 0
    for part in (kfs[:3], kfs[3:5], kfs[5:6], [], kfs[6:7]):   # batches of 3, 2, 1, none and 1 clouds
        for k in part:
            for o in oras:
                o.integrate(k["xyz"], rgba_of(k), k["Twc"])
        if part:
            xyz, rgba, offsets, Twc = _batch(part)
            single.integrate_batch_dev(xyz, rgba, offsets, Twc)
            want_visits = single.last_stats()["visits"]
        else:
            xyz, rgba, offsets, Twc = (torch.zeros((0, 3), device="cuda"), torch.zeros((0, 4), dtype=torch.uint8, device="cuda"),
                                       np.zeros(1, np.int32), torch.zeros((0, 3, 4), device="cuda"))
            want_visits = 0
        counts = sharded_step(ranks, xyz, rgba, offsets, Twc)
        total = int(sum(c.sum() for c in counts))
        assert total == want_visits, "every voxel visit is cast by exactly one rank"
        assert sum(t.last_stats()["visits"] for t in ranks) == want_visits, "and applied by exactly one"
        if part and world > 1:
            idle = [r for r in range(world) if r >= len(part)]
            assert all(counts[r].sum() == 0 for r in idle), "a rank without a cloud of the call sends nothing"
        sent += total
    assert sent > 0
    blocks = 0
    for t, o in zip(ranks, oras):
        blocks += compare(o, t)
    assert blocks == compare_union(single, ranks)
    for t in ranks + [single]:
        t.close()


@pytest.mark.gpu
def test_hip_ray_sharded_voxblox_at_configs3_size_equals_the_single_device_layer():
    """2 cm voxels, full-resolution key frames with depths to 8 m (configs[3]): millions of visit records per call — the
    partition, the cloud sort and the voxel sort take the large-array radix passes — on three ranks; the union of the
    shards is the single-device layer bit for bit."""
    from plvs_amd.tsdf import TsdfVoxblox
    kfs = make_keyframes(5, room_size=(16.0, 12.0, 3.0), max_depth=8.0, seed=41)
    world = 3
    ranks = [TsdfVoxblox(0.02, max_blocks=65536, shard_rank=r, shard_count=world) for r in range(world)]
    single = TsdfVoxblox(0.02, max_blocks=65536)
    for part in (kfs[:4], kfs[4:]):
        xyz, rgba, offsets, Twc = _batch(part)
        single.integrate_batch_dev(xyz, rgba, offsets, Twc)
        counts = sharded_step(ranks, xyz, rgba, offsets, Twc)
        assert int(sum(c.sum() for c in counts)) == single.last_stats()["visits"]
        largest = max(locals().get("largest", 0), single.last_stats()["visits"])
    assert largest > (1 << 20), "the first call must take the large-array radix passes"
    assert compare_union(single, ranks) > 200
    for t in ranks + [single]:
        t.close()


def compare_union(single, ranks):
    """The shards together are the single-device layer: every block on exactly one rank, bit for bit."""
    ids = {tuple(x) for x in single.chunk_ids()}
    seen = {}
    for r, t in enumerate(ranks):
        for bid in (tuple(x) for x in t.chunk_ids()):
            assert bid not in seen, "a block lives on one rank"
            seen[bid] = r
    assert set(seen) == ids
    for bid in ids:
        a, b = single.get_chunk(*bid), ranks[seen[bid]].get_chunk(*bid)
        assert all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                  y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b)), bid
    return len(ids)


# ------------------------------------------------------------------ the orchestration over gloo (CPU)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeVoxblox:
    """Stands in for a rank's TsdfVoxblox in plvs_amd.shard.sharded_integrate_voxblox (CPU tensors): the "walk" makes
    (rank + 2 d + 1) records for destination d, each stamped (source, destination, index, 9); the "apply" keeps them."""

    def __init__(self, rank, world):
        self.rank, self.world, self.packed, self.applied = rank, world, False, None

    def _counts(self):
        return np.array([self.rank + 2 * d + 1 for d in range(self.world)], np.int64)

    def shard_walk(self, d_xyz, offsets, d_Twc):
        return self._counts()

    def shard_pack(self, send):
        row = 0
        for d in range(self.world):
            for i in range(int(self._counts()[d])):
                send[row] = torch.tensor([self.rank, d, i, 9], dtype=torch.int32)
                row += 1
        assert row == send.shape[0]
        self.packed = True

    def shard_apply(self, recv, counts, d_xyz, d_rgba, offsets, d_Twc):
        assert self.packed
        self.applied = (recv.clone(), [int(c) for c in counts])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    t = _FakeVoxblox(rank, world)
    timings = {}
    ok = True
    for _ in range(2):
        t.packed = False
        counts = shard_mod.sharded_integrate_voxblox(t, torch.zeros((4, 3)), None, np.array([0, 4], np.int32), None, timings=timings)
        ok &= np.array_equal(counts, t._counts())
        recv, rc = t.applied
        want = []
        for src in range(world):
            n = src + 2 * rank + 1
            ok &= rc[src] == n
            want += [[src, rank, i, 9] for i in range(n)]
        ok &= recv.tolist() == want
    ok &= sorted(timings) == ["apply", "exchange", "pack", "walk"]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_voxblox_integrate_orchestration(world):
    """plvs_amd.shard.sharded_integrate_voxblox end to end over gloo with a stand-in map: walk -> pack -> the all-to-all
    of the visit records -> apply, twice in a row, with the per-phase timings."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [(r, True) for r in range(world)]
