#!/usr/bin/env python3
"""Round 5: the streaming headline through the depth-image entry point (2-D tiles of 32 x 16 grid pixels) against the
point-stream entry point (tiles of 512 consecutive points), on two maps fed the same key frames: stage times per step,
tiles / deferred / runs, and the two maps compared bit for bit at the end.
  python scripts/experiments/depth_entry_stream.py [steps=8] [kf_per_step=100]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
from tests.synth_scene import make_keyframes, make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
KF = int(sys.argv[2]) if len(sys.argv) > 2 else 100
skf = make_stream_keyframes(NS * KF, threads=32, images=True)


def pack_cloud(kfs):
    return (torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
            torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
            np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


def pack_depth(kfs, step=2):
    gh, gw = kfs[0]["depth_grid"].shape
    d = torch.zeros((len(kfs), gh * step, gw * step), dtype=torch.float32, device="cuda")
    c = torch.zeros((len(kfs), gh * step, gw * step, 3), dtype=torch.uint8, device="cuda")
    d[:, ::step, ::step] = torch.from_numpy(np.stack([k["depth_grid"] for k in kfs])).cuda()
    c[:, ::step, ::step] = torch.from_numpy(np.stack([k["rgb_grid"] for k in kfs])).cuda()
    return (d, c, torch.from_numpy(kfs[0]["cam_grid"]).cuda(), step, 0.1, 5.0,
            torch.from_numpy(np.array([int(k["kfid"][0]) if len(k["kfid"]) else 0 for k in kfs], np.int32)).cuda(),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


clouds = [pack_cloud(skf[i * KF:(i + 1) * KF]) for i in range(NS)]
depths = [pack_depth(skf[i * KF:(i + 1) * KF]) for i in range(NS)]
room = None      # "room" / "room:parts": the saturated small room of the steady-state leg, depth input
os.environ.setdefault("PLVS_HIP_TSDF_TRACE", "0")
res = {}
maps = {}
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["cloud", "depth", "cloud", "depth"]
for name in MODES:
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    if ":" in name:      # "depth:64,64" = apply parts of 64 segments for chunks of more than 64
        name, parts = name.split(":")
        t.set_apply_parts(*[int(x) for x in parts.split(".")])
        print("parts", parts, end=" ")
    if name == "room":
        if room is None:
            room = pack_depth(make_keyframes(100, images=True))
        for _ in range(10):
            t.integrate_depth_batch_dev(*room)
        t.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = 0
        for _ in range(10):
            t.integrate_depth_batch_dev(*room)
            v += t.last_stats()["visits"]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        sm, c = t.stage_ms()
        print("room ms/step", round(dt * 1e3, 3), "visits/step", v // 10, {k: round(x / c, 4) for k, x in sm.items()},
              "sum", round(sum(sm.values()) / c, 4), flush=True)
        t.close()
        continue
    warm = min(3, NS - 1)
    for i in range(warm):
        (t.integrate_batch_dev(*clouds[i]) if name == "cloud" else t.integrate_depth_batch_dev(*depths[i]))
    t.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    visits = 0
    for i in range(warm, NS):
        (t.integrate_batch_dev(*clouds[i]) if name == "cloud" else t.integrate_depth_batch_dev(*depths[i]))
        visits += t.last_stats()["visits"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (NS - warm)
    sm, c = t.stage_ms()
    print(name, "ms/step", round(dt * 1e3, 3), "visits/step", visits // (NS - warm), {k: round(v / c, 4) for k, v in sm.items()},
          "sum", round(sum(sm.values()) / c, 4), flush=True)
    if name in maps:
        t.close()
    else:
        maps[name] = t
if len(maps) < 2:
    sys.exit(0)
a, b = maps["cloud"], maps["depth"]
ia, ib = {tuple(x) for x in a.chunk_ids()}, {tuple(x) for x in b.chunk_ids()}
bad = 0
if ia != ib:
    print("CHUNK SETS DIFFER", len(ia), len(ib))
    bad += 1
for cid in sorted(ia & ib):
    ca, cb = a.get_chunk(*cid), b.get_chunk(*cid)
    for k, nm in enumerate(("sdf", "weight", "kfid", "rgbw")):
        if not np.array_equal(ca[k].view(np.uint32), cb[k].view(np.uint32)):
            bad += 1
            if bad < 8:
                d = np.nonzero(ca[k].view(np.uint32) != cb[k].view(np.uint32))[0]
                print("DIFF", cid, nm, len(d), d[:4], ca[k].view(np.uint32)[d[:4]], cb[k].view(np.uint32)[d[:4]])
print("maps identical" if bad == 0 else f"maps DIFFER in {bad} planes", "chunks", len(ia), flush=True)
