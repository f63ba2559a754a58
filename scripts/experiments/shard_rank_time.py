#!/usr/bin/env python3
"""Per-rank GPU time of the ray-sharded chisel integrate (order-free mode), emulated on ONE device with
virtual ranks: N handles step through the bench's 100-keyframe batches, the all-to-all is done with tensor
slices (not timed: the driver measures the real multi-GPU runs).  Printed per N: the slowest rank's time in
shard_walk, shard_pack and shard_apply (HIP events around each phase), the bytes a rank sends, and what the
single-device integrate takes on the same batches."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402
from tests.test_shard_rays import WIDTHS, send_buffers, virtual_all_to_all  # noqa: E402

STEPS = 3
WEAK = "--weak" in sys.argv          # a step carries 100 keyframes per rank (bench.py's default at N > 1)
WORLDS = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [2, 4, 8]
PER_STEP = 100 * (max(WORLDS) if WEAK else 1)
poses = make_keyframes(100, max_depth=5.0, seed=0)
kfs = [poses[i % 100] for i in range(PER_STEP * STEPS)]


def batch(i):
    part = kfs[PER_STEP * i:PER_STEP * (i + 1)]
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in part])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in part])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in part]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in part])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in part]).astype(np.int32)
    return xyz, rgb, kfid, offsets, Twc


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b)


batches = [batch(i) for i in range(STEPS)]
single = TsdfChisel(0.05, max_chunks=16384, order_free=True)
t1 = []
for xyz, rgb, kfid, offsets, Twc in batches:
    _, ms = timed(lambda: single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc))
    t1.append(ms)
print("single device: %.3f ms per batch of %d keyframes (last %d of %d)" % (np.mean(t1[1:]), PER_STEP, STEPS - 1, STEPS))
single.close()
for world in WORLDS:
    ranks = [TsdfChisel(0.05, max_chunks=16384, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    per_step = []
    for xyz, rgb, kfid, offsets, Twc in batches:
        tw, tp, ta, sent = [], [], [], []
        counts, bufs = [], []
        for t in ranks:
            c, ms = timed(lambda: t.shard_walk(xyz, offsets, Twc))
            counts.append(c)
            tw.append(ms)
        for r, (t, c) in enumerate(zip(ranks, counts)):
            b, ms = timed(lambda: send_buffers(t, c))
            tp.append(ms)
            bufs.append(b)
            sent.append(sum(4 * WIDTHS[k] * (c[:, k].sum() - c[r, k]) for k in range(3)))
        for t, (seg, rec, run, rc) in zip(ranks, virtual_all_to_all(counts, bufs)):
            _, ms = timed(lambda: t.shard_apply(seg, rec, run, rc, rgb, kfid))
            ta.append(ms)
        sat = [t.shard_saturated() for t in ranks]
        for t in ranks:
            for lst in sat:
                if lst.shape[0]:
                    t.shard_note_saturated(lst)
        per_step.append((max(tw), max(tp), max(ta), max(sent), [t.last_stats()["visits"] for t in ranks]))
    w, p, a, sent, visits = (np.mean([s[k] for s in per_step[1:]]) if k < 4 else per_step[-1][4] for k in range(5))
    print("N=%d: slowest rank walk %.3f + pack %.3f + apply %.3f = %.3f ms per batch (speed-up of the compute %.2fx), "
          "%.1f MB sent per rank, visits per rank %d..%d"
          % (world, w, p, a, w + p + a, np.mean(t1[1:]) / (w + p + a), sent / 1e6, min(visits), max(visits)))
    for t in ranks:
        t.close()
