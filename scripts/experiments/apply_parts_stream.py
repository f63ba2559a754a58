#!/usr/bin/env python3
"""Stage times of the order-free step on the STREAMING workload of bench.py (steps of 100 distinct key frames of the
office loop) for several (part_segments, min_segments) of the apply stage; the saturated small room beside it."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes, make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def pack(kfs):
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


skf = make_stream_keyframes(NS * 100, threads=32)
steps = [pack(skf[i * 100:(i + 1) * 100]) for i in range(NS)]
room = pack(make_keyframes(100, max_depth=5.0, seed=0))
settings = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or \
    [(256, 2048), (256, 512), (128, 256), (64, 128), (128, 128), (64, 64), (512, 1024)]
for ps, pm in settings:
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    t.set_apply_parts(ps, pm)
    for b in steps[:3]:
        t.integrate_batch_dev(*b)
    t.set_profiling(True)
    for b in steps[3:]:
        t.integrate_batch_dev(*b)
    sm, n = t.stage_ms()
    t.set_profiling(False)
    t.close()
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    t.set_apply_parts(ps, pm)
    for _ in range(8):
        t.integrate_batch_dev(*room)
    t.set_profiling(True)
    for _ in range(8):
        t.integrate_batch_dev(*room)
    rm, rn = t.stage_ms()
    t.close()
    print(ps, pm, "stream", {k: round(v / n, 4) for k, v in sm.items()}, "sum", round(sum(sm.values()) / n, 4),
          "| room", {k: round(v / rn, 4) for k, v in rm.items()}, "sum", round(sum(rm.values()) / rn, 4), flush=True)
