#!/usr/bin/env python3
"""Phase clocks of apply_chunks (developer build: make variant NAME=prof DEFS=-DPLVS_WALK_PROF) on the streaming workload
and in the saturated room: shader cycles of thread 0 of every workgroup per phase, items, longest item."""
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


def pack_depth(kfs, step=2):      # (round 5: the depth-image entry point, PLVS_APPLY_PROF_DEPTH=1)
    gh, gw = kfs[0]["depth_grid"].shape
    d = torch.zeros((len(kfs), gh * step, gw * step), dtype=torch.float32, device="cuda")
    c = torch.zeros((len(kfs), gh * step, gw * step, 3), dtype=torch.uint8, device="cuda")
    d[:, ::step, ::step] = torch.from_numpy(np.stack([k["depth_grid"] for k in kfs])).cuda()
    c[:, ::step, ::step] = torch.from_numpy(np.stack([k["rgb_grid"] for k in kfs])).cuda()
    return (d, c, torch.from_numpy(kfs[0]["cam_grid"]).cuda(), step, 0.1, 5.0,
            torch.from_numpy(np.array([int(k["kfid"][0]) if len(k["kfid"]) else 0 for k in kfs], np.int32)).cuda(),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


DEPTH = os.environ.get("PLVS_APPLY_PROF_DEPTH", "0") == "1"
if DEPTH:
    pack = pack_depth
NS = 10
skf = make_stream_keyframes(NS * 100, threads=32, images=DEPTH)
steps = [pack(skf[i * 100:(i + 1) * 100]) for i in range(NS)]
room = [pack(make_keyframes(100, max_depth=5.0, seed=0, images=DEPTH))] * 10
NAMES = ["set-up", "segments(thread 0)", "wait slowest wave", "part merge", "voxel updates"]
PARTS = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(256, 2048)]
for name, seq, (ps, pm) in [(n, q, p) for p in PARTS for n, q in (("stream", steps), ("room", room))]:
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    t.set_apply_parts(ps, pm)
    print((ps, pm), end=" ")
    for b in seq[:4]:
        (t.integrate_depth_batch_dev if DEPTH else t.integrate_batch_dev)(*b)
    _lib.lib.plvs_hip_debug_walk_prof(None, 1)
    t.set_profiling(True)
    for b in seq[4:]:
        (t.integrate_depth_batch_dev if DEPTH else t.integrate_batch_dev)(*b)
    sm, c = t.stage_ms()
    buf = (ctypes.c_ulonglong * 16)()
    _lib.lib.plvs_hip_debug_walk_prof(buf, 0)
    tot = float(sum(buf[9:14])) or 1.0
    print(name, {k: round(v / c, 4) for k, v in sm.items()},
          {NAMES[i]: round(buf[9 + i] / tot, 3) for i in range(5)},
          "items/call", buf[14] // c, "cycles/item", int(tot / max(buf[14], 1)), "longest item: cycles", buf[15] >> 24, "segments", (buf[15] >> 4) & 0xFFFFF,
          "parts of its chunk", buf[15] & 15, flush=True)
    if hasattr(_lib.lib, "plvs_hip_debug_apply_items"):      # the items of the LAST call, longest first
        n_items = int(min(buf[14] // max(c, 1) + 64, 8192))
        it = (ctypes.c_ulonglong * (4 * n_items))()
        _lib.lib.plvs_hip_debug_apply_items(it, n_items)
        a = np.frombuffer(it, np.uint64).reshape(-1, 4)
        order = np.argsort(-a[:, 0].astype(np.int64))[:8]
        print("   longest items (cycles, segments, records of group 0, first-16 rounds, cycles waiting for their loads, cycles in their adds):",
              [(int(a[i, 0]), int(a[i, 1] >> np.uint64(32)), int(a[i, 1] & np.uint64(0xFFFFFFFF)), int(a[i, 3] & np.uint64(255)),
                int(a[i, 2]), int(a[i, 3] >> np.uint64(8))) for i in order], flush=True)
        cyc = np.sort(a[:, 0].astype(np.int64))[::-1]
        print("   item cycles: max", int(cyc[0]), "10th", int(cyc[9]), "100th", int(cyc[99]), "median", int(np.median(cyc[cyc > 0])), flush=True)
    t.close()
