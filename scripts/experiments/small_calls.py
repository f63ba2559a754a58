"""The order-free integrate in calls of K key frames (default 5: PointCloudMapping::UpdateMap's batch) on a fresh map
(first lap) and on the map that lap left (second lap): wall time per call and the library's stage times.
usage: small_calls.py [K] [laps] [--ordered]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

ORDERED = "--ordered" in sys.argv
ARGS = [a for a in sys.argv[1:] if not a.startswith("-")]
K = int(ARGS[0]) if len(ARGS) > 0 else 5
LAPS = int(ARGS[1]) if len(ARGS) > 1 else 2
kfs = make_keyframes(100, max_depth=5.0, seed=0)
calls = []
for j0 in range(0, 100, K):
    g = kfs[j0:j0 + K]
    calls.append((torch.from_numpy(np.concatenate([k["xyz"] for k in g])).cuda(),
                  torch.from_numpy(np.concatenate([k["rgb"] for k in g])).cuda(),
                  torch.from_numpy(np.concatenate([k["kfid"] for k in g]).astype(np.int32)).cuda(),
                  np.cumsum([0] + [k["xyz"].shape[0] for k in g]).astype(np.int32),
                  torch.from_numpy(np.stack([k["Twc"] for k in g])).cuda()))
t = TsdfChisel(0.05, max_chunks=16384, order_free=not ORDERED)
for c in calls:           # warm-up lap: sizes the scratch buffers
    t.integrate_batch_dev(*c)
t.clear()
for lap in range(LAPS):
    t.set_profiling(True)
    ts, visits = [], 0
    for c in calls:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.integrate_batch_dev(*c)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        visits += t.last_stats()["visits"]
    sm, n = t.stage_ms()
    t.set_profiling(False)
    print(f"K={K} lap {lap + 1}: median {np.median(ts) * 1e3:.4f} ms  max {max(ts) * 1e3:.4f} ms  "
          f"{visits / sum(ts) / 1e9:.2f} Gvoxels/s  stages/call {({k: round(v / n, 4) for k, v in sm.items()})}")
t.close()
