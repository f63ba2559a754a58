"""Ray-sharded multi-GPU integrate of the chisel back end (plvs_hip_tsdf_chisel_shard_walk / _pack / _apply),
run as VIRTUAL ranks on one device: N handles, the all-to-all emulated with tensor slices.  The partial sums are
integers, so the union of the shards must equal the single-device order-free map bit for bit — sdf, weight, kfid
and colour — for any N."""
import numpy as np
import pytest
import torch

from plvs_amd.shard import owner_of
from plvs_amd.synth_scene import make_keyframes


def _batch(kfs):
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    return xyz, rgb, kfid, offsets, Twc


def virtual_all_to_all(counts, segs, recs):
    """counts[src][dst] = (segments, records); buffers grouped by destination -> per destination, grouped by source."""
    world = len(counts)
    out = []
    for dst in range(world):
        ps, pr, rc = [], [], np.zeros((world, 2), np.int64)
        for src in range(world):
            so, ro = counts[src][:dst].sum(axis=0)
            ns, nr = counts[src][dst]
            ps.append(segs[src][so:so + ns])
            pr.append(recs[src][ro:ro + nr])
            rc[src] = (ns, nr)
        out.append((torch.cat(ps).contiguous(), torch.cat(pr).contiguous(), rc))
    return out


def sharded_step(ranks, xyz, rgb, kfid, offsets, Twc):
    counts = [t.shard_walk(xyz, offsets, Twc) for t in ranks]
    segs, recs = [], []
    for t, c in zip(ranks, counts):
        seg = torch.zeros((int(c[:, 0].sum()), 8), dtype=torch.int32, device="cuda")
        rec = torch.zeros((int(c[:, 1].sum()), 4), dtype=torch.int32, device="cuda")
        t.shard_pack(seg, rec)
        segs.append(seg)
        recs.append(rec)
    torch.cuda.synchronize()
    for t, (seg, rec, rc) in zip(ranks, virtual_all_to_all(counts, segs, recs)):
        t.shard_apply(seg, rec, rc, xyz, rgb, kfid)
    torch.cuda.synchronize()
    return counts


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_hip_ray_sharded_integrate_equals_the_single_device_map(world):
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(8, max_depth=5.0, seed=3)
    single = TsdfChisel(0.05, max_chunks=4096, order_free=True)
    ranks = [TsdfChisel(0.05, max_chunks=4096, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    walked = 0
    for b0 in range(0, len(kfs), 3):   # batches of 3, 3 and 2 keyframes: later calls meet half-saturated colours
        xyz, rgb, kfid, offsets, Twc = _batch(kfs[b0:b0 + 3])
        single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        want_visits = single.last_stats()["visits"]
        sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
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
        t.shard_pack(torch.zeros((0, 8), dtype=torch.int32, device="cuda"), torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
        t.shard_apply(torch.zeros((0, 8), dtype=torch.int32, device="cuda"), torch.zeros((0, 4), dtype=torch.int32, device="cuda"),
                      np.zeros((world, 2), np.int64), xyz[:0], rgb[:0], kfid[:0])
    for t in ranks + [single]:
        t.close()
