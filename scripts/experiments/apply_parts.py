#!/usr/bin/env python3
"""Stage times of the order-free step (100 key frames, steady state) for several (part_segments, min_segments) of the
apply stage."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
for ps, pm in ((256, 2048), (256, 512), (128, 256), (64, 128), (512, 1024), (128, 512)):
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    t.set_apply_parts(ps, pm)
    for _ in range(10):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    t.set_profiling(True)
    for _ in range(10):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    sm, n = t.stage_ms()
    print(ps, pm, {k: round(v / n, 4) for k, v in sm.items()}, "sum", round(sum(sm.values()) / n, 4), flush=True)
    t.close()
