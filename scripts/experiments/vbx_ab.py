#!/usr/bin/env python3
"""A/B of the two voxblox integrate pipelines (PLVS_VBX_SORTED=1 forces the sorting one): same clouds into two maps,
every voxel of every block compared bit for bit; wall time per call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfVoxblox  # noqa: E402

nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
res = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
kfs = make_keyframes(nkf, room_size=(16.0, 12.0, 3.0), max_depth=8.0, seed=0)
for k in kfs:
    k["rgba"] = np.concatenate([k["rgb"], np.full((k["rgb"].shape[0], 1), 255, np.uint8)], axis=1)
os.environ["PLVS_VBX_SORTED"] = "1"
a = TsdfVoxblox(res, max_blocks=65536)
del os.environ["PLVS_VBX_SORTED"]
b = TsdfVoxblox(res, max_blocks=65536)
for lap in range(2):
    for s in range(0, nkf, batch):
        sel = kfs[s:s + batch]
        xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in sel])).cuda()
        rgba = torch.from_numpy(np.concatenate([k["rgba"] for k in sel])).cuda()
        Twc = torch.from_numpy(np.stack([k["Twc"] for k in sel])).cuda()
        offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in sel]).astype(np.int32)
        ts = []
        for m in (a, b):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.integrate_batch_dev(xyz, rgba, offsets, Twc)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        sa, sb = a.last_stats(), b.last_stats()
        print(f"lap {lap} call {s}: sorted {ts[0]:.3f} ms, sort-free {ts[1]:.3f} ms | stats", sa, sb, flush=True)
ia = sorted(tuple(int(v) for v in x) for x in a.chunk_ids())
ib = sorted(tuple(int(v) for v in x) for x in b.chunk_ids())
print("blocks", len(ia), len(ib), "same ids", ia == ib)
bad = 0
for bid in ia:
    if bid not in set(ib):
        continue
    for name, x, y in zip(("distance", "weight", "colour"), a.get_chunk(*bid), b.get_chunk(*bid)):
        ne = int((x.view(np.uint32) != y.view(np.uint32)).sum())
        if ne:
            bad += 1
            if bad < 6:
                idx = np.flatnonzero(x.view(np.uint32).ravel() != y.view(np.uint32).ravel())[:4]
                print("mismatch", name, bid, ne, "voxels", idx, x.ravel()[idx], y.ravel()[idx])
print("mismatching planes:", bad)
