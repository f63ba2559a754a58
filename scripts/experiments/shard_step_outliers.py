#!/usr/bin/env python3
"""Per-step, per-phase wall times of the ray-sharded integrate at N = 1 on the streaming workload: which phase holds the
outliers (buffers that grow inside the job)?"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
from plvs_amd.shard import sharded_integrate  # noqa: E402
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dist.init_process_group("nccl", rank=0, world_size=1)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 14
skf = make_stream_keyframes(NS * 100, threads=32)


def pack(kfs):
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


steps = [pack(skf[i * 100:(i + 1) * 100]) for i in range(NS)]
t = TsdfChisel(0.05, max_chunks=16384, order_free=True, shard_rank=0, shard_count=1)
import time
for i, b in enumerate(steps):
    tim = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c = sharded_integrate(t, *b, timings=tim)
    torch.cuda.synchronize()
    print(i, round((time.perf_counter() - t0) * 1e3, 2), {k: round(v, 2) for k, v in tim.items()},
          "descriptors / voxel sums / runs sent:", c.sum(axis=0).tolist(), flush=True)
for i, b in enumerate(steps[:6]):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sharded_integrate(t, *b)
    torch.cuda.synchronize()
    print("untimed", i, round((time.perf_counter() - t0) * 1e3, 2), flush=True)

# ---- where the host time of a bench step goes: the pieces of bench.py's `step` at N = 1
from plvs_amd.shard import BlockDirectory, allgather_block_lists  # noqa: E402
d_upd = torch.zeros((16384, 3), dtype=torch.int32, device="cuda")
gdir = BlockDirectory(16384)
acc = {}
t2 = TsdfChisel(0.05, max_chunks=16384, order_free=True, shard_rank=0, shard_count=1)
for i, b in enumerate(steps):
    marks = [time.perf_counter()]
    sharded_integrate(t2, *b); marks.append(time.perf_counter())
    st = t2.last_stats(); marks.append(time.perf_counter())
    n = t2.updated_chunk_ids_dev(d_upd); marks.append(time.perf_counter())
    all_ids, counts = allgather_block_lists(d_upd, n, 16384, padded=True); marks.append(time.perf_counter())
    gdir.merge(all_ids, counts); marks.append(time.perf_counter())
    torch.cuda.synchronize(); marks.append(time.perf_counter())
    if i >= 4:
        for name, a, b_ in zip(("sharded_integrate", "last_stats", "updated_ids", "allgather", "merge", "final sync"), marks[:-1], marks[1:]):
            acc[name] = acc.get(name, 0.0) + (b_ - a) * 1e3 / (len(steps) - 4)
print("host pieces of a step, ms:", {k: round(v, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()), 3), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in steps[4:10]:
    sharded_integrate(t2, *b)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
