#!/usr/bin/env python3
"""ORB || lines through plvs_hip_frame_extract_dev: wall time per frame and the two extractors' own stage times."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd.frame import extract_frame  # noqa: E402
from plvs_amd.lines import LineExtractor  # noqa: E402
from plvs_amd.orb import ORBextractor  # noqa: E402
from tests.pgm import golden_frame as golden  # noqa: E402

frames = [torch.from_numpy(golden(n)).cuda() for n in ("aloe_640x480.pgm", "aloe_640x480_shift.pgm", "cones_640x480.pgm")]
ext, lext = ORBextractor(2000, 1.2, 8, 20, 7), LineExtractor(100)
for i in range(8):
    extract_frame(ext, lext, frames[i % 3])
so, sl = {}, {}
t0 = time.perf_counter()
N = 60
for i in range(N):
    extract_frame(ext, lext, frames[i % 3])
    for k, v in ext.stage_ms().items():
        so[k] = so.get(k, 0.0) + v
    for k, v in lext.stage_ms().items():
        sl[k] = sl.get(k, 0.0) + v
ms = (time.perf_counter() - t0) / N * 1e3
print(f"{ms:.3f} ms per frame | orb", {k: round(v / N, 3) for k, v in so.items()}, "sum", round(sum(so.values()) / N, 3),
      "| lines", {k: round(v / N, 3) for k, v in sl.items()})
