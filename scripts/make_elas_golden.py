#!/usr/bin/env python3
"""Writes tests/golden/elas_capture.npz: the arguments the reference's libelas pipeline hands to Elas::computeDisparity and
Elas::adaptiveMean — and what the reference's own compiled methods return — for a 256 x 128 crop (at x 330, y 170) of the tree's urban1 pair
(tests/golden/urban1*) with PLVS's postprocess_only_left: both computeDisparity calls at full resolution, the adaptiveMean call
at full resolution and with subsampling.  Run where
oracle/_ref/libelas_ref.so exists (the dev container); tests/test_elas.py replays the capture against the oracle (CPU) and
the HIP path (GPU) where the compiled reference is absent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import elas_ref  # noqa: E402
from tests.test_elas import pair  # noqa: E402

out = {}
nd = nm = 0
left, right = pair()
left, right = np.ascontiguousarray(left[170:298, 330:586]), np.ascontiguousarray(right[170:298, 330:586])
for sub in (False, True):
    disp_calls, mean_calls, _ = elas_ref.capture(left, right, subsampling=sub, plvs=True)
    for a in disp_calls if not sub else []:      # (the 16-byte descriptor images make a call 1 MB: full resolution only)
        for k in ("grid", "grid_dims", "D"):
            out[f"d{nd}_{k}"] = a[k]
        if nd == 0:
            out["I1_desc"], out["I2_desc"] = a["I1_desc"], a["I2_desc"]
        else:
            assert np.array_equal(out["I1_desc"], a["I1_desc"]) and np.array_equal(out["I2_desc"], a["I2_desc"])
        out[f"d{nd}_support"] = a["support"].view(np.int32).reshape(-1, 3)
        out[f"d{nd}_tri"] = a["tri"].view(np.uint32).reshape(-1, 9)
        for k in ("right_image", "width", "height", "subsampling"):
            out[f"d{nd}_{k}"] = np.int32(a[k])
        nd += 1
    for m in mean_calls:
        out[f"m{nm}_D_in"], out[f"m{nm}_D_out"] = m["D_in"], m["D_out"]
        for k in ("width", "height", "subsampling"):
            out[f"m{nm}_{k}"] = np.int32(m[k])
        nm += 1
# the candidate grid of computeSupportMatches for the stored descriptor pair: the reference keeps it in a local variable,
# so the stored grid is the oracle's — AFTER checking that the reference pipeline fed with it hands computeDisparity the
# very support points and triangles of the pure reference run (the capture above)
from tests import oracle_lib  # noqa: E402
ora = oracle_lib.load()
grids = []
seen = []
elas_ref.run_with(left, right, lambda a: (seen.append(a), ora.elas_compute_disparity(a))[1], ora.elas_adaptive_mean,
                  subsampling=False, plvs=True,
                  support_candidates=lambda a: (grids.append(ora.elas_support_candidates(a)), grids[-1])[1])
ref_calls, _, _ = elas_ref.capture(left, right, subsampling=False, plvs=True)
assert len(grids) == 1 and all(np.array_equal(g["support"], w["support"]) and np.array_equal(g["tri"], w["tri"])
                               for g, w in zip(seen, ref_calls))
out["D_can"] = grids[0]
out["n_disparity_calls"], out["n_mean_calls"] = np.int32(nd), np.int32(nm)
path = os.path.join(ROOT, "tests", "golden", "elas_capture.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", nd, "disparity calls,", nm, "mean calls")
