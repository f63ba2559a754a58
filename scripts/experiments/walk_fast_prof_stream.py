#!/usr/bin/env python3
"""Phase clocks of walk_fast (developer build: make variant NAME=prof DEFS=-DPLVS_WALK_PROF) on the streaming workload and in
the saturated room: shader cycles of thread 0 of every tile between the tile's barriers."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
os.environ.setdefault("PLVS_HIP_LIB", os.path.join(ROOT, "plvs_amd", "lib", "libplvs_hip_prof.so"))
from plvs_amd import _lib  # noqa: E402
from tests.synth_scene import make_keyframes, make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402


def pack(kfs):
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


NS = 10
skf = make_stream_keyframes(NS * 100, threads=32)
steps = [pack(skf[i * 100:(i + 1) * 100]) for i in range(NS)]
room = [pack(make_keyframes(100, max_depth=5.0, seed=0))] * 10
NAMES = ["set-up", "wait 1", "voxel loop (wave 0)", "wait slowest wave", "entries: chunks, ranks, colour weights", "wait 3", "records",
         "runs", "epilogue"]
for name, seq in (("stream", steps), ("room", room)):
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    for b in seq[:4]:
        t.integrate_batch_dev(*b)
    _lib.lib.plvs_hip_debug_walk_prof(None, 1)
    t.set_profiling(True)
    for b in seq[4:]:
        t.integrate_batch_dev(*b)
    sm, c = t.stage_ms()
    buf = (ctypes.c_ulonglong * 16)()
    _lib.lib.plvs_hip_debug_walk_prof(buf, 0)
    tot = float(sum(buf[0:9])) or 1.0
    print(name, {k: round(v / c, 4) for k, v in sm.items()}, "cycles per call (thread 0 of all tiles)", int(tot / c), flush=True)
    for i in range(9):
        print(f"    {NAMES[i]:40s} {buf[i] / tot:.3f}", flush=True)
    t.close()
