#!/usr/bin/env python3
"""What would tiles of spatially close rays be worth?  The order-free step on the bench stream and on the saturated room with
every cloud's points (a) as they come (row-major over the image grid), (b) re-ordered on the HOST into blocks of the viewing
directions (x/z, y/z binned, blocks in row-major order, the original order kept inside a block) — the kernels unchanged: a tile
is still 512 consecutive points of a cloud, only now they are neighbours in two dimensions.  Prints stage times and last_stats."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes, make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def reorder(k, bw, bh, delta):
    if bw == 0:
        return k
    x, y, z = k["xyz"][:, 0], k["xyz"][:, 1], k["xyz"][:, 2]
    zz = np.where(z > 1e-6, z, 1.0)
    cx = np.clip(np.floor(x / zz / delta).astype(np.int64) + 512, 0, 1023)
    cy = np.clip(np.floor(y / zz / delta).astype(np.int64) + 512, 0, 1023)
    key = (cy // bh) * 1024 + cx // bw
    o = np.argsort(key, kind="stable")
    return {"xyz": k["xyz"][o], "rgb": k["rgb"][o], "kfid": k["kfid"][o], "Twc": k["Twc"]}


def pack(kfs):
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


skf = make_stream_keyframes(NS * 100, threads=32)
rkf = make_keyframes(100, max_depth=5.0, seed=0)
# delta: one bin = one grid point of the 320x240 grid at f = 262.5 (the bench clouds); blocks in grid points
D = 1.0 / 262.5
for bw, bh in [(0, 0), (32, 16), (16, 32), (23, 23), (64, 8), (16, 16), (45, 12)]:
    steps = [pack([reorder(k, bw, bh, D) for k in skf[i * 100:(i + 1) * 100]]) for i in range(NS)]
    room = pack([reorder(k, bw, bh, D) for k in rkf])
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    for b in steps[:3]:
        t.integrate_batch_dev(*b)
    t.set_profiling(True)
    for b in steps[3:]:
        t.integrate_batch_dev(*b)
    sm, n = t.stage_ms()
    st = t.last_stats()
    t.close()
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    for _ in range(8):
        t.integrate_batch_dev(*room)
    t.set_profiling(True)
    for _ in range(8):
        t.integrate_batch_dev(*room)
    rm, rn = t.stage_ms()
    t.close()
    print((bw, bh), "stream", {k: round(v / n, 4) for k, v in sm.items()}, "sum", round(sum(sm.values()) / n, 4),
          "| room", {k: round(v / rn, 4) for k, v in rm.items()}, "sum", round(sum(rm.values()) / rn, 4), flush=True)
    print("     ", st, flush=True)
