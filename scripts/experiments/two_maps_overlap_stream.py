#!/usr/bin/env python3
"""two_maps_overlap.py on the STREAMING workload: two independent maps integrate the same 12 steps of 100 distinct key
frames from two host threads on two streams; the aggregate time per call against one map alone bounds what pipelining
consecutive calls of ONE map (the tail of call k — segment sort, apply, colour chain — under the walk of call k + 1)
could gain."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

NS = 12
skf = make_stream_keyframes(NS * 100, threads=32)
steps = []
for i in range(NS):
    kfs = skf[i * 100:(i + 1) * 100]
    steps.append((torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda(),
                  torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda(),
                  torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda(),
                  np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32),
                  torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()))


def run(t, stream, lo, hi):
    with torch.cuda.stream(stream):
        for b in steps[lo:hi]:
            t.integrate_batch_dev(*b)
    stream.synchronize()


for rep in range(2):
    maps = [TsdfChisel(0.05, max_chunks=16384, order_free=True) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    for t, s in zip(maps, streams):
        run(t, s, 0, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(maps[0], streams[0], 3, NS)
    one = (time.perf_counter() - t0) * 1e3 / (NS - 3)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(t, s, 3, NS)) for t, s in zip(maps[1:], streams[1:])]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) * 1e3 / (2 * (NS - 3))
    print("one map: %.3f ms per call; two maps concurrently: %.3f ms per call (aggregate) -> %.1f %% less" % (one, two, 100 * (1 - two / one)), flush=True)
    for t in maps:
        t.close()
