"""Front-end time per frame (points and lines on two threads) against the host thread knobs.
Run on the GPU box:  PYTHONPATH=. python scripts/experiments/frontend_threads.py"""
import os
import time

import numpy as np
import torch

from plvs_amd.frame import extract_frame
from plvs_amd.lines import LineExtractor
from plvs_amd.orb import ORBextractor
from plvs_amd.pgm import golden_frame as golden


def measure(nrep=60):
    frames = [torch.from_numpy(golden(n)).cuda() for n in ("aloe_640x480.pgm", "aloe_640x480_shift.pgm",
                                                            "cones_640x480.pgm")]
    ext = ORBextractor(2000, 1.2, 8, 20, 7)
    lext = LineExtractor(100)
    for i in range(6):
        extract_frame(ext, lext, frames[i % 3])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(nrep):
            extract_frame(ext, lext, frames[i % 3])
        best = min(best, (time.perf_counter() - t0) / nrep * 1e3)
    t0 = time.perf_counter()
    for i in range(nrep):
        ext(frames[i % 3])
    orb = (time.perf_counter() - t0) / nrep * 1e3
    t0 = time.perf_counter()
    for i in range(nrep):
        lext(frames[i % 3])
    lines = (time.perf_counter() - t0) / nrep * 1e3
    ext.close()
    lext.close()
    return best, orb, lines


if __name__ == "__main__":
    print("host cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)))
    for tree in (8, 4, 3, 2, 1):
        for fit, yld in ((4, 0), (4, 1), (2, 0), (2, 1), (1, 1), (0, 0)):
            os.environ["PLVS_HIP_ORB_TREE_THREADS"] = str(tree)
            os.environ["PLVS_HIP_LINES_FIT_THREADS"] = str(fit)
            os.environ["PLVS_HIP_LINES_FIT_YIELD"] = str(yld)
            both, orb, lines = measure()
            print(f"tree {tree} fit {fit} yield {yld}: both {both:.3f} ms  orb {orb:.3f}  lines {lines:.3f}", flush=True)
