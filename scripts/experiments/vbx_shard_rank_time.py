#!/usr/bin/env python3
"""Per-rank GPU time of the ray-sharded voxblox integrate ("simple", 2 cm, the office stream), emulated on ONE device with
virtual ranks: N handles step through batches of the stream, the all-to-all is done with tensor slices (not timed: the driver
measures the real multi-GPU runs).  Printed per N: the slowest rank's time in shard_walk, shard_pack and shard_apply (HIP events
around each phase), the bytes a rank sends, and what the single-device integrate takes on the same batches.
usage: vbx_shard_rank_time.py [--weak] [N ...]   (--weak: a step carries 25 key frames PER RANK; default: 25 per step)"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfVoxblox  # noqa: E402

STEPS = 3
WEAK = "--weak" in sys.argv
WORLDS = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [2, 4, 8]
PER_STEP = 25 * (max(WORLDS) if WEAK else 1)
kfs = make_stream_keyframes(PER_STEP * STEPS, first=400, max_depth=8.0, seed=0, threads=32)


def batch(i):
    part = kfs[PER_STEP * i:PER_STEP * (i + 1)]
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in part])).cuda(),
            torch.from_numpy(np.concatenate([np.concatenate([k["rgb"], np.full((k["rgb"].shape[0], 1), 255, np.uint8)], axis=1) for k in part])).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in part]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in part])).cuda())


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b)


batches = [batch(i) for i in range(STEPS)]
single = TsdfVoxblox(0.02, max_blocks=65536)
t1 = []
for xyz, rgba, offsets, Twc in batches:
    _, ms = timed(lambda: single.integrate_batch_dev(xyz, rgba, offsets, Twc))
    t1.append(ms)
print("single device: %.3f ms per batch of %d key frames (last %d of %d)" % (np.mean(t1[1:]), PER_STEP, STEPS - 1, STEPS))
single.close()
for world in WORLDS:
    ranks = [TsdfVoxblox(0.02, max_blocks=65536, shard_rank=r, shard_count=world) for r in range(world)]
    per_step = []
    for xyz, rgba, offsets, Twc in batches:
        tw, tp, ta, sent, counts, sends = [], [], [], [], [], []
        for t in ranks:
            c, ms = timed(lambda: t.shard_walk(xyz, offsets, Twc))
            counts.append(c)
            tw.append(ms)
        for r, (t, c) in enumerate(zip(ranks, counts)):
            buf = torch.empty((int(c.sum()), 4), dtype=torch.int32, device="cuda")
            _, ms = timed(lambda: t.shard_pack(buf))
            sends.append(buf)
            tp.append(ms)
            sent.append(16 * (int(c.sum()) - int(c[r])))
        for dst, t in enumerate(ranks):
            parts, rc = [], np.zeros(world, np.int64)
            for src in range(world):
                off = int(counts[src][:dst].sum())
                rc[src] = counts[src][dst]
                parts.append(sends[src][off:off + int(rc[src])])
            recv = torch.cat(parts).contiguous()
            _, ms = timed(lambda: t.shard_apply(recv, rc, xyz, rgba, offsets, Twc))
            ta.append(ms)
        per_step.append((max(tw), max(tp), max(ta), max(sent)))
    w, p, a, s = (np.mean([x[k] for x in per_step[1:]]) for k in range(4))
    print("N = %d: slowest rank walk %.3f + pack %.3f + apply %.3f = %.3f ms per step (%.2f x the single device's), %.1f MB sent per rank"
          % (world, w, p, a, w + p + a, np.mean(t1[1:]) / (w + p + a), s / 1e6))
    for t in ranks:
        t.close()
