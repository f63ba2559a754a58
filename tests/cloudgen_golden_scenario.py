"""The depth image -> cloud golden scenario.  Run three ways over the SAME code below:
  scripts/make_cloudgen_golden.py            PointCloudMapping::InitCamGridPoints / ::GeneratePointCloudInCameraFrameBGRA cut
                                             verbatim out of the reference's src/PointCloudMapping.cc and compiled here
                                             (oracle/_ref/libcloudgen_ref.so) -> tests/golden/cloudgen_reference_digests.json
  tests/test_oracle_pinned_cloudgen.py, CPU  oracle/cloudgen.c reproduces the file (no oracle/_ref needed)
  tests/test_oracle_pinned_cloudgen.py, GPU  the HIP path, through the C ABI, reproduces the file
A generator is  make(width, height, step, K (3x3 f32), min_depth, max_depth) -> (grid [n, 2] f32,
                gen(depth [h, w] f32, bgr [h, w, 3] u8, kfid) -> (records as bytes-compatible [n] of 48 B, pixel_to_point [h, w] i32))."""
import hashlib

import numpy as np

from tests.plvs_amd_synth import TUM1, make_rgbd_frames

K_TUM = np.array([[TUM1["fx"], 0, TUM1["cx"]], [0, TUM1["fy"], TUM1["cy"]], [0, 0, 1]], np.float32)
CASES = [
    dict(id="tum_step1", width=640, height=480, step=1, seed=1, min_depth=0.1, max_depth=5.0),     # (first: the largest grid)
    dict(id="tum_step2", width=640, height=480, step=2, seed=0, min_depth=0.1, max_depth=5.0),
    dict(id="odd_step3", width=637, height=479, step=3, seed=2, min_depth=0.1, max_depth=5.0),
    dict(id="near_far_limits", width=600, height=400, step=2, seed=3, min_depth=1.5, max_depth=3.25),
    dict(id="step4_small", width=162, height=122, step=4, seed=4, min_depth=0.01, max_depth=10.0),
]


def _sha(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def frames(case, n=2):
    return [(np.ascontiguousarray(f["depth"][:case["height"], :case["width"]]),
             np.ascontiguousarray(f["bgr"][:case["height"], :case["width"]])) for f in make_rgbd_frames(n, seed=case["seed"], holes=True)]


def run(make):
    out = {}
    for case in CASES:
        grid, gen = make(case["width"], case["height"], case["step"], K_TUM, case["min_depth"], case["max_depth"])
        rec = dict(grid=_sha(grid))
        for k, (depth, bgr) in enumerate(frames(case)):
            pts, p2p = gen(depth, bgr, 40 + k)
            rec[f"frame{k}"] = dict(n=int(len(pts)), points=_sha(pts), pixel_to_point=_sha(p2p))
        out[case["id"]] = rec
    return out
