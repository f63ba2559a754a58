#!/usr/bin/env python3
"""Where a ray-sharded step's wall time goes on one rank (world 1 over RCCL: the collectives are there, the wire
is not): walk / pack / exchange / apply / saturation feedback / block lists, each bracketed by a device sync."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd import shard  # noqa: E402
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
t = TsdfChisel(0.05, max_chunks=16384, shard_rank=0, shard_count=1, order_free=True)
d_upd = torch.zeros((16384, 3), dtype=torch.int32, device="cuda")
acc = {}


def lap(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


STEPS = 12
for s in range(STEPS + 2):
    if s == 2:
        acc.clear()
    counts = lap("walk", lambda: t.shard_walk(xyz, offsets, Twc))
    seg = torch.empty((int(counts[:, 0].sum()), 8), dtype=torch.int32, device="cuda")
    rec = torch.empty((int(counts[:, 1].sum()), 8), dtype=torch.int32, device="cuda")
    run = torch.empty((int(counts[:, 2].sum()), 6), dtype=torch.int32, device="cuda")
    lap("pack", lambda: t.shard_pack(seg, rec, run))
    rs, rr, ru, rc = lap("exchange", lambda: shard.exchange_segments(seg, rec, run, counts))
    lap("apply", lambda: t.shard_apply(rs, rr, ru, rc, rgb, kfid))
    sat = lap("saturated", lambda: t.shard_saturated())
    n = lap("updated ids", lambda: t.updated_chunk_ids_dev(d_upd))
    lap("block lists", lambda: shard.allgather_block_lists(d_upd, n, 16384, padded=True))
whole = 0.0
for s in range(STEPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    shard.sharded_integrate(t, xyz, rgb, kfid, offsets, Twc)
    n = t.updated_chunk_ids_dev(d_upd)
    shard.allgather_block_lists(d_upd, n, 16384, padded=True)
    torch.cuda.synchronize()
    whole += (time.perf_counter() - t0) * 1e3
print("per step, ms:", {k: round(v / STEPS, 3) for k, v in acc.items()}, "| sum", round(sum(acc.values()) / STEPS, 3),
      "| one sharded_integrate + block lists without the extra syncs", round(whole / STEPS, 3))
dist.destroy_process_group()
