#!/usr/bin/env python3
"""How much do two order-free integrate pipelines overlap on one GPU?  Two maps, two host threads, two streams, each fed the
same stream of 100-key-frame steps: the time of both together against twice the time of one — an upper bound on what a step
could gain from running its apply / colour stages beside the next step's walk."""
import sys
import threading
import time

import numpy as np
import torch

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

NS, KF = 16, 100
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


depths = [pack_depth(skf[i * KF:(i + 1) * KF]) for i in range(NS)]


def run(t, stream, first, bar):
    with torch.cuda.stream(stream):
        for i in range(3):
            t.integrate_depth_batch_dev(*depths[i])
        bar.wait()
        for i in range(3, NS):
            t.integrate_depth_batch_dev(*depths[i])
        stream.synchronize()


for nthreads in (1, 2, 3):
    maps = [TsdfChisel(0.05, max_chunks=16384, order_free=True) for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    bar = threading.Barrier(nthreads + 1)
    th = [threading.Thread(target=run, args=(maps[k], streams[k], 0, bar)) for k in range(nthreads)]
    for x in th:
        x.start()
    bar.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} pipelines: {dt * 1e3 / (NS - 3):.3f} ms per step of all, {dt * 1e3 / (NS - 3) / nthreads:.3f} ms per step per pipeline", flush=True)
    for m in maps:
        m.close()
