#!/usr/bin/env python3
"""Per-rank GPU time of the sharded chisel integrate, emulated on ONE device: rank 0 of N = 1, 2, 4, 8
on the bench's 100-keyframe batch (no collective).  The driver measures the real multi-GPU runs; this
shows what a rank computes."""
import time

import numpy as np
import torch

from plvs_amd.synth_scene import make_keyframes
from plvs_amd.tsdf import TsdfChisel

kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
for n in (1, 2, 4, 8):
    t = TsdfChisel(0.05, max_chunks=16384, shard_rank=0, shard_count=n)
    for _ in range(2):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    t.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st, calls = t.stage_ms()
    print(n, "ranks: %.3f ms per step on rank 0, visits %d" % (dt * 1e3, t.last_stats()["visits"]),
          {k: round(v / max(calls, 1), 3) for k, v in st.items()})
    t.close()
