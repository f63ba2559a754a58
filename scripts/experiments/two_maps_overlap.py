#!/usr/bin/env python3
"""How much of an integrate call's tail (segment sort, apply, colour fold) can hide behind another call's walk?
Two independent maps integrate the bench batch from two host threads on two streams; aggregate throughput against
one map alone bounds what pipelining consecutive calls of ONE map could gain."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
STEPS = 10


def run(t, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    stream.synchronize()


maps = [TsdfChisel(0.05, max_chunks=16384, order_free=True) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for t, s in zip(maps, streams):
    run(t, s, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(maps[0], streams[0], STEPS)
one = (time.perf_counter() - t0) * 1e3 / STEPS
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(t, s, STEPS)) for t, s in zip(maps, streams)]
for x in th:
    x.start()
for x in th:
    x.join()
torch.cuda.synchronize()
two = (time.perf_counter() - t0) * 1e3 / (2 * STEPS)
print("one map: %.3f ms per call; two maps concurrently: %.3f ms per call (aggregate) -> %.1f %% less" % (one, two, 100 * (1 - two / one)))
