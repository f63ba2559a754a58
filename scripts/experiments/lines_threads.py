#!/usr/bin/env python3
"""Line extraction time per 640x480 frame for the PLVS_HIP_LINES_FIT_THREADS of the environment."""
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd.lines import LineExtractor  # noqa: E402
from tests.pgm import golden_frame as golden  # noqa: E402

frames = [torch.from_numpy(golden(n)).cuda() for n in ("aloe_640x480.pgm", "aloe_640x480_shift.pgm", "cones_640x480.pgm")]
lext = LineExtractor(100)
for i in range(6):
    lext(frames[i % 3])
st = {}
t0 = time.perf_counter()
for i in range(60):
    lext(frames[i % 3])
    for k, v in lext.stage_ms().items():
        st[k] = st.get(k, 0.0) + v
ms = (time.perf_counter() - t0) / 60 * 1e3
print(os.environ.get("PLVS_HIP_LINES_FIT_THREADS", "default"), f"{ms:.3f} ms", {k: round(v / 60, 3) for k, v in st.items()})
