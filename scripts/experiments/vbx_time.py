#!/usr/bin/env python3
"""Wall time of voxblox integrate calls (25 key frames each, second lap) for the library PLVS_HIP_LIB names."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfVoxblox  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 25
kfs = make_keyframes(50, room_size=(16.0, 12.0, 3.0), max_depth=8.0, seed=0)
for k in kfs:
    k["rgba"] = np.concatenate([k["rgb"], np.full((k["rgb"].shape[0], 1), 255, np.uint8)], axis=1)
b = TsdfVoxblox(0.02, max_blocks=65536)
batches = []
for s in range(0, 50, 25):
    sel = kfs[s:s + NB]
    batches.append((torch.from_numpy(np.concatenate([k["xyz"] for k in sel])).cuda(),
                    torch.from_numpy(np.concatenate([k["rgba"] for k in sel])).cuda(),
                    np.cumsum([0] + [k["xyz"].shape[0] for k in sel]).astype(np.int32),
                    torch.from_numpy(np.stack([k["Twc"] for k in sel])).cuda()))
for lap in range(3):
    for xyz, rgba, offsets, Twc in batches:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.integrate_batch_dev(xyz, rgba, offsets, Twc)
        torch.cuda.synchronize()
        if lap:
            print(f"{(time.perf_counter() - t0) * 1e3:.3f} ms", b.last_stats(), flush=True)
