#!/usr/bin/env python3
"""Where the waves of vb_chain_chunks spend their time (a developer build named by PLVS_HIP_LIB:
`make -C plvs_amd/csrc variant NAME=prof UNIT=tsdf_voxblox DEFS=-DPLVS_VB_PROF=1`): per call, the kernel's span, the mean time
a wave spends on the head list / the short runs / the long runs, the longest wave and when it started.  (The finer probes
of round 3 — per-trip phases, group-trips per wave — were taken out of the kernel again; DESIGN §4.1 has what they showed.)"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd import _lib  # noqa: E402
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfVoxblox  # noqa: E402

lib = ctypes.CDLL(_lib.LIB_PATH)
kfs = make_keyframes(50, room_size=(16.0, 12.0, 3.0), max_depth=8.0, seed=0)
for k in kfs:
    k["rgba"] = np.concatenate([k["rgb"], np.full((k["rgb"].shape[0], 1), 255, np.uint8)], axis=1)
b = TsdfVoxblox(0.02, max_blocks=65536)
batches = []
for s in range(0, 50, 25):
    sel = kfs[s:s + 25]
    batches.append((torch.from_numpy(np.concatenate([k["xyz"] for k in sel])).cuda(),
                    torch.from_numpy(np.concatenate([k["rgba"] for k in sel])).cuda(),
                    np.cumsum([0] + [k["xyz"].shape[0] for k in sel]).astype(np.int32),
                    torch.from_numpy(np.stack([k["Twc"] for k in sel])).cuda()))
WPB = int(sys.argv[1]) if len(sys.argv) > 1 else 4   # waves per workgroup of the build
for lap in range(3):
    for xyz, rgba, offsets, Twc in batches:
        b.integrate_batch_dev(xyz, rgba, offsets, Twc)
        nw = min(-(-b.last_stats()["visits"] // 2048) * WPB, 1 << 16)
        buf = np.zeros((nw, 4), np.uint64)       # per wave: the times it started, finished the head list, the short runs, the kernel
        lib.plvs_hip_debug_chain_prof(buf.ctypes.data_as(ctypes.c_void_p), nw)
        if lap:
            t = buf.astype(np.int64)
            t0 = t[:, 0].min()
            d = np.diff(t, axis=1) / 100.0                      # 100 MHz ticks -> us
            tot = (t[:, 3] - t[:, 0]) / 100.0
            i = int(np.argmax(tot))
            print(f"waves {nw}  kernel span {(t[:, 3].max() - t0) / 100.0:.1f} us;  mean per wave: heads {d[:, 0].mean():.2f} short {d[:, 1].mean():.2f} "
                  f"long {d[:, 2].mean():.2f} us;  longest wave {tot[i]:.1f} us (heads {d[i, 0]:.1f} short {d[i, 1]:.1f} long {d[i, 2]:.1f}), "
                  f"started at {(t[i, 0] - t0) / 100.0:.1f} us;  last wave start {(t[:, 0].max() - t0) / 100.0:.1f} us;  "
                  f"waves longer than 50 us: {(tot > 50).sum()}, 100 us: {(tot > 100).sum()}", flush=True)
