#!/usr/bin/env python3
"""Phase clocks of walk_fast on the headline's input (depth images of the office stream), developer build
`make -C plvs_amd/csrc variant NAME=prof DEFS=-DPLVS_WALK_PROF`: shader cycles of thread 0 of every tile between the tile's
barriers, and the stage times of the call.  Usage (GPU box): python scripts/experiments/r6_walk_prof_depth.py [steps]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
if os.path.exists(os.path.join(ROOT, "plvs_amd", "lib", "libplvs_hip_prof.so")) and "PLVS_HIP_LIB" not in os.environ:
    os.environ["PLVS_HIP_LIB"] = os.path.join(ROOT, "plvs_amd", "lib", "libplvs_hip_prof.so")
from plvs_amd import _lib  # noqa: E402
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402


def pack_depth(sel, step=2, max_depth=5.0):
    gh, gw = sel[0]["depth_grid"].shape
    d = torch.zeros((len(sel), gh * step, gw * step), dtype=torch.float32, device="cuda")
    c = torch.zeros((len(sel), gh * step, gw * step, 3), dtype=torch.uint8, device="cuda")
    d[:, ::step, ::step] = torch.from_numpy(np.stack([k["depth_grid"] for k in sel])).cuda()
    c[:, ::step, ::step] = torch.from_numpy(np.stack([k["rgb_grid"] for k in sel])).cuda()
    return (d, c, torch.from_numpy(sel[0]["cam_grid"]).cuda(), step, 0.1, max_depth,
            torch.from_numpy(np.array([int(k["kfid"][0]) if len(k["kfid"]) else 0 for k in sel], np.int32)).cuda(),
            torch.from_numpy(np.stack([k["Twc"] for k in sel])).cuda())


NS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(os.environ.get("BATCH", "100"))
skf = make_stream_keyframes(NS * B, threads=32, images=True, max_depth=5.0, seed=0)
steps = [pack_depth(skf[i * B:(i + 1) * B]) for i in range(NS)]
NAMES = ["set-up", "wait 1", "voxel loop (wave 0)", "wait slowest wave", "entries: chunks, ranks, colour weights", "wait 3", "records",
         "runs", "epilogue"]
t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
for b in steps[:4]:
    t.integrate_depth_batch_dev(*b)
has_prof = hasattr(_lib.lib, "plvs_hip_debug_walk_prof")
if has_prof:
    _lib.lib.plvs_hip_debug_walk_prof(None, 1)
t.set_profiling(True)
for b in steps[4:]:
    t.integrate_depth_batch_dev(*b)
    print("   stats", t.last_stats(), flush=True)
sm, c = t.stage_ms()
print("stage ms per call", {k: round(v / c, 4) for k, v in sm.items()}, "calls", c, flush=True)
if has_prof:
    buf = (ctypes.c_ulonglong * 16)()
    _lib.lib.plvs_hip_debug_walk_prof(buf, 0)
    tot = float(sum(buf[0:9])) or 1.0
    print("cycles per call (thread 0 of all tiles)", int(tot / c), flush=True)
    for i in range(9):
        print(f"    {NAMES[i]:40s} {buf[i] / tot:.3f}", flush=True)
t.close()
