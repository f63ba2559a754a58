#!/usr/bin/env python3
"""GPU time of the dense stereo (SGM) call on the KITTI-sized pair, images resident in HBM."""
import numpy as np
import torch

from tests.pgm import golden_frame as golden
from plvs_amd.sgm import StereoSGM

left = np.ascontiguousarray(golden("urban1_1241x376.pgm")[:, :1240])
right = np.ascontiguousarray(golden("urban1_right_1241x376.pgm")[:, :1240])
h, w = left.shape
sgm = StereoSGM(w, h)
dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
out = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
for _ in range(3):
    sgm.execute_dev(dl, dr, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    sgm.execute_dev(dl, dr, out)
e1.record()
torch.cuda.synchronize()
print("sgm %dx%d: %.3f ms per pair, valid %.2f" % (w, h, e0.elapsed_time(e1) / 20, float((out > 0).float().mean())))
