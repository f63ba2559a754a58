#!/usr/bin/env python3
"""What would a step gain if its later images knew which voxels the earlier ones had saturated?  Every 100-key-frame step of the
stream as ONE call against the same step as TWO calls (the first `first` key frames, then the rest): stage times, tiles and runs
of each call.  The two-call form pays a call's fixed latencies twice — the question is what the SECOND call's walk and colour
chain cost once the first call's colours are in the map.   python scripts/experiments/split_step_probe.py [first=25]"""
import os
import sys

import numpy as np
import torch

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 25
NS, KF = 14, 100
skf = make_stream_keyframes(NS * KF, threads=32, images=True)


def pack_depth(kfs, step=2):
    gh, gw = kfs[0]["depth_grid"].shape
    d = torch.zeros((len(kfs), gh * step, gw * step), dtype=torch.float32, device="cuda")
    c = torch.zeros((len(kfs), gh * step, gw * step, 3), dtype=torch.uint8, device="cuda")
    d[:, ::step, ::step] = torch.from_numpy(np.stack([k["depth_grid"] for k in kfs])).cuda()
    c[:, ::step, ::step] = torch.from_numpy(np.stack([k["rgb_grid"] for k in kfs])).cuda()
    return (d, c, torch.from_numpy(kfs[0]["cam_grid"]).cuda(), step, 0.1, 5.0,
            torch.from_numpy(np.array([int(k["kfid"][0]) if len(k["kfid"]) else 0 for k in kfs], np.int32)).cuda(),
            torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda())


def run(parts):
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    calls = []
    for i in range(NS):
        lo = i * KF
        for a, b in parts:
            calls.append((i, pack_depth(skf[lo + a:lo + b])))
    t.set_profiling(True)
    acc = {}
    for i, args in calls:
        if i == 3:
            torch.cuda.synchronize()
            t.stage_ms()       # (reads and leaves the counters: the difference below)
            base = {k: v for k, v in t.stage_ms()[0].items()}
            n0 = t.stage_ms()[1]
        t.integrate_depth_batch_dev(*args)
    torch.cuda.synchronize()
    sm, c = t.stage_ms()
    steps = NS - 3
    print(parts, {k: round((sm[k] - base[k]) / steps, 4) for k in sm}, "sum per step", round(sum(sm[k] - base[k] for k in sm) / steps, 4),
          flush=True)
    t.close()


run([(0, 100)])
run([(0, FIRST), (FIRST, 100)])
run([(0, 10), (10, 100)])
run([(0, 50), (50, 100)])
