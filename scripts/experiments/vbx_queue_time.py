#!/usr/bin/env python3
"""voxblox InsertCloud per key frame (host flavour): one integrate per key frame against queue x 5 + flush (the mirrors'
UpdateMap batch), 2 cm, the bench's stream."""
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from tests.synth_scene import make_stream_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfVoxblox  # noqa: E402

kfs = make_stream_keyframes(60, first=400, max_depth=8.0, threads=16)
for k in kfs:
    k["rgba"] = np.concatenate([k["rgb"], np.full((len(k["rgb"]), 1), 255, np.uint8)], 1)
for mode in ("one by one", "queue 5 + flush"):
    t = TsdfVoxblox(0.02, max_blocks=65536)
    for k in kfs[:10]:
        t.integrate(k["xyz"], k["rgba"], k["Twc"])
    t0 = time.perf_counter()
    if mode == "one by one":
        for k in kfs[10:]:
            t.integrate(k["xyz"], k["rgba"], k["Twc"])
    else:
        for j in range(10, 60, 5):
            for k in kfs[j:j + 5]:
                t.queue(k["xyz"], k["rgba"], k["Twc"])
            t.flush()
    dt = time.perf_counter() - t0
    print(mode, round(dt / 50 * 1e3, 3), "ms per key frame (incl. the host -> device copy of the cloud)", t.num_chunks(), "blocks")
    t.close()
