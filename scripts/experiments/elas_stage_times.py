#!/usr/bin/env python3
"""Wall time of each device stage of libelas (host-pointer entry points) on the urban1 pair; needs oracle/_ref."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd.elas import ElasGPU  # noqa: E402
from tests.pgm import golden_frame as golden  # noqa: E402
from tests import elas_ref  # noqa: E402

el, er = golden("urban1_1241x376.pgm"), golden("urban1_right_1241x376.pgm")
for sub in (False, True):
    dcalls, mcalls, ref_out = elas_ref.capture(el, er, subsampling=sub, plvs=True)
    e = ElasGPU(ElasGPU.Parameters(subsampling=sub))
    d0 = dcalls[0]
    w, h = d0["width"], d0["height"]
    t = {}

    def timed(name, f, *a):
        f(*a)
        t0 = time.perf_counter()
        for _ in range(10):
            out = f(*a)
        t[name] = (time.perf_counter() - t0) / 10 * 1e3
        return out
    timed("descriptors", e.setImages, el, er)
    d1, d2 = e.descriptors()
    assert np.array_equal(d1, d0["I1_desc"]) and np.array_equal(d2, d0["I2_desc"])
    timed("candidates", e.supportCandidates, None, None, w, h)
    Ds = [timed(f"disparity{i}", e.computeDisparity, a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, a["right_image"], w, h)
          for i, a in enumerate(dcalls)]
    D1, D2 = timed("lr", e.leftRightConsistencyCheck, Ds[0], Ds[1], w, h)
    D1 = timed("segments", e.removeSmallSegments, D1, w, h)
    D1 = timed("gaps", e.gapInterpolation, D1, w, h)
    D1 = timed("mean", e.adaptiveMean, D1, w, h)
    print("subsampling" if sub else "full", {k: round(v, 3) for k, v in t.items()}, "identical", np.array_equal(D1, ref_out[0]))
