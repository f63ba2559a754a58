"""N > 1 path on CPU: world_size 2 and 3 gloo runs of the block-list exchange, with the
CPU oracle standing in for each rank's shard of the map.

Checks: the owner function used by the host equals the one compiled into the
oracle/kernels (disjoint shards whose union is the unsharded map), and the
all-gather returns every rank's updated-block list to every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # plvs_amd/shard.py needs numpy + torch only: loaded by path, so that a box without the HIP library (the
    # package import loads it) still runs this CPU test
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    allgather_block_lists, owner_of = shard_mod.allgather_block_lists, shard_mod.owner_of
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes, TUM1
    cam = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        cam[k] /= 4
    cam["width"] //= 4
    cam["height"] //= 4
    oracle = oracle_lib.load()
    shard = oracle.chisel(0.05, shard_rank=rank, shard_count=world)
    seen = []
    cap = 512
    for kf in make_keyframes(2, cam=cam, seed=5):
        before = {tuple(x) for x in shard.chunk_ids()}
        shard.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        new = sorted({tuple(x) for x in shard.chunk_ids()} - before)
        buf = torch.zeros((cap, 3), dtype=torch.int32)
        if new:
            buf[: len(new)] = torch.tensor(new, dtype=torch.int32)
        lists = allgather_block_lists(buf, len(new), cap)
        assert len(lists) == world
        assert [tuple(x) for x in lists[rank].tolist()] == new
        for r, l in enumerate(lists):
            if len(l):
                assert np.all(owner_of(l.numpy(), world) == r)      # host owner fn == kernel/oracle owner fn
        seen.append([[tuple(x) for x in l.tolist()] for l in lists])
    q.put((rank, seen, sorted(tuple(x) for x in shard.chunk_ids())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])      # 3: the owner function's non-power-of-two path
def test_gloo_block_list_exchange(world):
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
    seen = [o[1] for o in out]
    ids = [set(o[2]) for o in out]
    assert all(s == seen[0] for s in seen)                 # every rank saw the same gathered lists
    for a in range(world):
        for b in range(a + 1, world):
            assert not (ids[a] & ids[b])                   # shards are disjoint
    sys.path.insert(0, ROOT)
    from tests import oracle_lib
    from tests.plvs_amd_synth import make_keyframes, TUM1
    cam = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        cam[k] /= 4
    cam["width"] //= 4
    cam["height"] //= 4
    full = oracle_lib.load().chisel(0.05)
    for kf in make_keyframes(2, cam=cam, seed=5):
        full.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    union = set().union(*ids)
    assert union == {tuple(x) for x in full.chunk_ids()}
    gathered = {t for step in seen[0] for l in step for t in l}
    assert gathered == union                               # the exchange announced every new block


def _worker_segments(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    # rank `src` sends to rank `dst`: (src + 2 dst) segments, (3 src + dst) records, (src * dst) runs, every item
    # stamped with (src, dst, index) so that the receiver can tell where each row comes from
    counts = np.array([[rank + 2 * d, 3 * rank + d, rank * d] for d in range(world)], np.int64)
    bufs = []
    for k, width in enumerate((8, 8, 6)):
        rows = []
        for d in range(world):
            for i in range(counts[d, k]):
                rows.append([rank, d, i, k] + [7] * (width - 4))
        bufs.append(torch.tensor(rows, dtype=torch.int32).reshape(-1, width))
    seg, rec, run, rc = shard_mod.exchange_segments(bufs[0], bufs[1], bufs[2], counts)
    ok = True
    for k, (got, width) in enumerate(((seg, 8), (rec, 8), (run, 6))):
        want = []
        for src in range(world):
            n = [src + 2 * rank, 3 * src + rank, src * rank][k]
            ok &= int(rc[src, k]) == n
            want += [[src, rank, i, k] + [7] * (width - 4) for i in range(n)]
        ok &= got.tolist() == want
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_segment_exchange_of_the_ray_sharded_integrate(world):
    """plvs_amd.shard.exchange_segments: counts + three all_to_all_single with split sizes (empty splits included)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [(r, True) for r in range(world)]


class _FakeIntegrate:
    """Stands in for a rank's TsdfChisel in plvs_amd.shard.sharded_integrate (CPU tensors): the "walk" produces, for
    every destination rank d, (rank + 2 d) descriptors, (3 rank + d + 1) voxel sums and (rank * d) runs, every row
    stamped (source, destination, index, kind); the "apply" keeps what arrived; every rank reports (rank + 1) voxels as
    saturated after the apply, and a rank notes what the others (and itself) reported."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.applied, self.noted, self.packed = None, [], False

    def _counts(self):
        return np.array([[self.rank + 2 * d, 3 * self.rank + d + 1, self.rank * d] for d in range(self.world)], np.int64)

    def shard_walk(self, d_xyz, offsets, d_Twc):
        return self._counts()

    def shard_pack(self, seg, rec, run):
        for k, buf in enumerate((seg, rec, run)):
            row = 0
            for d in range(self.world):
                for i in range(int(self._counts()[d, k])):
                    buf[row, :4] = torch.tensor([self.rank, d, i, k], dtype=torch.int32)
                    buf[row, 4:] = 7
                    row += 1
            assert row == buf.shape[0]
        self.packed = True

    def shard_apply(self, seg, rec, run, counts, d_rgb, d_kfid):
        assert self.packed
        self.applied = (seg.clone(), rec.clone(), run.clone(), np.array(counts))

    def shard_saturated_message(self, msg, rows):
        mine = torch.tensor([[self.rank, 0, 0, v] for v in range(self.rank + 1)], dtype=torch.int32).reshape(-1, 4)
        msg[: mine.shape[0]] = mine
        msg[rows] = torch.tensor([mine.shape[0], 0, 0, 0], dtype=torch.int32)

    def shard_note_gathered(self, gathered, nranks, rows):
        g = gathered.view(nranks, rows + 1, 4)
        for r in range(nranks):
            self.noted.append(g[r, : int(g[r, rows, 0])].clone())


def _worker_sharded_integrate(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    t = _FakeIntegrate(rank, world)
    xyz = torch.zeros((4, 3))
    timings = {}
    ok = True
    for step in range(2):      # (the second step also takes the path with the first one's timings present)
        t.noted, t.packed = [], False
        counts = shard_mod.sharded_integrate(t, xyz, None, None, np.array([0, 4], np.int32), None, timings=timings)
        ok &= np.array_equal(counts, t._counts())
        seg, rec, run, rc = t.applied
        for k, (got, width) in enumerate(((seg, 8), (rec, 8), (run, 6))):
            want = []
            for src in range(world):
                n = [src + 2 * rank, 3 * src + rank + 1, src * rank][k]
                ok &= int(rc[src, k]) == n
                want += [[src, rank, i, k] + [7] * (width - 4) for i in range(n)]
            ok &= got.tolist() == want
        # every rank's saturated voxels reached this rank, in rank order
        noted = [v.tolist() for v in t.noted]
        ok &= noted == [[[src, 0, 0, v] for v in range(src + 1)] for src in range(world)]
    ok &= sorted(timings) == ["apply", "exchange", "feedback", "pack", "walk"] and all(v >= 0.0 for v in timings.values())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_integrate_orchestration(world):
    """plvs_amd.shard.sharded_integrate end to end over gloo with a stand-in map: walk -> pack -> the all-to-all of
    the three buffers -> apply -> the all-gather of the saturated voxels -> note, twice in a row, with the per-phase
    timings bench.py --gpus N reports."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded_integrate, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [(r, True) for r in range(world)]


class _FakeShard:
    """Stands in for a rank's TsdfChisel in the meshing-halo protocol (CPU tensors): the map is a line of chunks
    (i, 0, 0), i < n, hash-sharded like the real one; "meshing" chunk i needs chunk i + 1, and — once that one is
    here and its payload is odd — chunk i + 2 as well (new vertices reaching further: a second round).  Chunks past
    the end of the line do not exist anywhere."""
    HALO_WORDS = 8
    device = torch.device("cpu")

    def __init__(self, rank, world, n, owner_of):
        self.rank, self.world, self.n = rank, world, n
        ids = np.array([[i, 0, 0] for i in range(n)], np.int32)
        self.owner = {i: int(o) for i, o in enumerate(owner_of(ids, world))}
        self.own = {i: self.payload_of(i) for i in range(n) if self.owner[i] == rank}
        self.ghost, self.absent, self.missing = {}, set(), []

    @staticmethod
    def payload_of(i):
        return [(i * 7 + 3) % 11] + [i] * 7

    def _get(self, i, miss):
        if i in self.own:
            return self.own[i]
        if i in self.ghost:
            return self.ghost[i]
        owner = self.owner.get(i)
        if owner is None:                 # beyond the line: owned by nobody this rank knows; ask the hash owner
            if i in self.absent:
                return None
            miss.add(i)
            return None
        if owner == self.rank or i in self.absent:
            return None
        miss.add(i)
        return None

    def mesh_chunks(self, chunk_ids, halo_ok=False):
        miss, out = set(), {}
        for c in chunk_ids:
            i = int(c[0])
            if i not in self.own:
                continue
            acc = list(self.own[i])
            nb = self._get(i + 1, miss)
            if nb is not None:
                acc.append(nb[0])
                if nb[0] % 2 == 1:
                    nb2 = self._get(i + 2, miss)
                    if nb2 is not None:
                        acc.append(nb2[0])
            out[i] = acc
        self.missing = sorted(miss)
        return None if self.missing else out

    def mesh_probe(self, chunk_ids):
        self.mesh_chunks(chunk_ids)
        return len(self.missing)

    def halo_missing(self):
        return np.array([[i, 0, 0] for i in self.missing], np.int32).reshape(-1, 3)

    def halo_lookup(self, d_ids, found):
        for k, c in enumerate(d_ids.tolist()):
            found[k] = 1 if c[0] in self.own else 0

    def halo_export(self, d_ids, found, payload):
        row = 0
        for k, c in enumerate(d_ids.tolist()):
            if int(found[k]):
                payload[row] = torch.tensor(self.own[c[0]], dtype=torch.int32)
                row += 1
        assert row == payload.shape[0]

    def halo_import(self, d_ids, found, payload):
        row = 0
        for k, c in enumerate(d_ids.tolist()):
            if int(found[k]):
                self.ghost[c[0]] = payload[row].tolist()
                row += 1
            else:
                self.absent.add(c[0])
        assert row == payload.shape[0]


def _worker_halo(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    n = 23
    fake = _FakeShard(rank, world, n, shard_mod.owner_of)
    fake.owner[n] = int(shard_mod.owner_of(np.array([[n, 0, 0]]), world)[0])        # ids past the end: asked of the
    fake.owner[n + 1] = int(shard_mod.owner_of(np.array([[n + 1, 0, 0]]), world)[0])  # rank the hash names, "absent"
    mine = np.array([[i, 0, 0] for i in sorted(fake.own)], np.int32).reshape(-1, 3)
    got, fetched = shard_mod.sharded_mesh_chunks(fake, mine)
    q.put((rank, got, fetched, sorted(fake.ghost), sorted(fake.absent)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_meshing_halo_rounds(world):
    """plvs_amd.shard.sharded_mesh_chunks / halo_round over gloo with a stand-in map: requests reach the owners,
    answers (found and not found) come back to the asker in the asker's order, a second round fetches what the first
    round's chunks made reachable, ranks that are done keep serving, and the result equals the unsharded one."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_halo, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 23
    pay = _FakeShard.payload_of
    want = {}
    for i in range(n):
        acc = list(pay(i))
        if i + 1 < n:
            acc.append(pay(i + 1)[0])
            if pay(i + 1)[0] % 2 == 1 and i + 2 < n:
                acc.append(pay(i + 2)[0])
        want[i] = acc
    merged = {}
    for rank, got, fetched, ghosts, absent in out:
        assert not (set(got) & set(merged))
        merged.update(got)
        assert fetched == len(ghosts)
    assert merged == want
    assert any(o[2] > 0 for o in out) and any(o[4] for o in out), "chunks were fetched, and some were reported absent"


def test_voxblox_halo_ids_are_the_foreign_forward_neighbours():
    """plvs_amd.shard.voxblox_halo_ids: exactly the +x / +y / +z neighbours (seven per block) owned by other ranks,
    once each."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("plvs_amd_shard", os.path.join(ROOT, "plvs_amd", "shard.py"))
    shard_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard_mod)
    rng = np.random.default_rng(3)
    ids = np.unique(rng.integers(-6, 6, (200, 3)), axis=0).astype(np.int32)
    for world in (2, 3, 8):
        for rank in range(world):
            mine = ids[shard_mod.owner_of(ids, world) == rank]
            need = shard_mod.voxblox_halo_ids(mine, world, rank)
            want = set()
            for b in mine:
                for dx in (0, 1):
                    for dy in (0, 1):
                        for dz in (0, 1):
                            if dx or dy or dz:
                                n = (int(b[0]) + dx, int(b[1]) + dy, int(b[2]) + dz)
                                if shard_mod.owner_of(np.array([n]), world)[0] != rank:
                                    want.add(n)
            assert len(need) == len(want) and set(map(tuple, need.tolist())) == want
    assert shard_mod.voxblox_halo_ids(np.zeros((0, 3), np.int32), 4, 0).shape == (0, 3)
