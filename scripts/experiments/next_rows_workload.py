#!/usr/bin/env python3
"""The rows next to the hot path at their working sizes, a few repetitions each — the command profiled into
profiles/r01_next_rows_kernel_stats.md:
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python scripts/experiments/next_rows_workload.py
Prints host-side time per call (synchronous entry points, results downloaded)."""
import time

import numpy as np
import torch

from plvs_amd import cloudgen
from plvs_amd.orb import ORBextractor
from tests.pgm import golden_frame as golden
from plvs_amd.sgm import StereoSGM
from plvs_amd.stereo import StereoMatcher
from tests.synth_scene import TUM1, make_keyframes, make_rgbd_frames
from plvs_amd.tsdf import TsdfChisel


def timed(name, fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call", flush=True)


# depth image -> cloud (640x480, stride 2: 76 800 grid points)
w, h = TUM1["width"], TUM1["height"]
grid = cloudgen.InitCamGridPoints(w, h, 2, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
gen = cloudgen.PointCloudGenerator(w, h, grid, step=2, min_depth=0.1, max_depth=5.0)
fr = make_rgbd_frames(1, seed=2)[0]
timed("cloudgen 640x480 step 2 (host in / host out)", lambda: gen.GeneratePointCloudInCameraFrameBGRA(fr["bgr"], fr["depth"], 7), 20)

# sparse stereo on the KITTI-sized pair (2000 features per image)
left, right = golden("urban1_1241x376.pgm"), golden("urban1_right_1241x376.pgm")
exl, exr = ORBextractor(2000, 1.2, 8, 20, 7), ORBextractor(2000, 1.2, 8, 20, 7)
_, kl, dl = exl(left)
_, kr, dr = exr(right)
sm = StereoMatcher(exl, exr)
bf = np.float32(386.1448)
mb = np.float32(bf / np.float32(718.856))
timed("stereo matches 2000 x 2000 keypoints, 1241x376", lambda: sm.ComputeStereoMatches(kl, dl, kr, dr, mb, bf), 20)

# dense stereo
l8, r8 = np.ascontiguousarray(left[:, :1240]), np.ascontiguousarray(right[:, :1240])
sgm = StereoSGM(1240, 376)
d_l, d_r = torch.from_numpy(l8).cuda(), torch.from_numpy(r8).cuda()
out = torch.zeros((376, 1240), dtype=torch.uint8, device="cuda")
timed("sgm 1240x376, 64 disparities (device in / device out)", lambda: sgm.execute_dev(d_l, d_r, out), 20)

# meshes of the whole 100-keyframe map
m = TsdfChisel(0.05, max_chunks=16384)
for k in make_keyframes(100, seed=0):
    m.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
ids = np.ascontiguousarray(m.chunk_ids(), np.int32)
res = {}
def mesh():
    res["m"] = m.mesh_chunks(ids)
timed(f"mesh {len(ids)} chunks", mesh, 5, warm=1)
print("vertices:", len(res["m"]["vertices"]))
