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
out["n_disparity_calls"], out["n_mean_calls"] = np.int32(nd), np.int32(nm)
path = os.path.join(ROOT, "tests", "golden", "elas_capture.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", nd, "disparity calls,", nm, "mean calls")
