import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from plvs_amd.orb import ORBextractor
from plvs_amd.pgm import golden_frame as golden
from plvs_amd.orbmatcher import FrameView, LastFrameView, ORBmatcher
ext = ORBextractor(2000, 1.2, 8, 20, 7)
scale = np.asarray(ext.GetScaleFactors(), np.float32)
f0, f1 = (torch.from_numpy(golden(n)).cuda() for n in ("aloe_640x480.pgm", "aloe_640x480_shift.pgm"))
_, k0, d0 = ext(f0)
_, k1, d1 = ext(f1)
last = LastFrameView(valid=np.ones(len(k0), np.uint8), u=k0["x"] - 3.0, v=k0["y"] - 2.0, invz=np.full(len(k0), 0.5, np.float32), octave=k0["octave"], angle=k0["angle"], desc=d0)
cur = FrameView(k1["x"], k1["y"], k1["octave"], np.full(len(k1), -1.0, np.float32), d1, 0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0, scale)
om = ORBmatcher(0.9, True)
for _ in range(3):
    t0 = time.perf_counter()
    n, a = om.SearchByProjectionLastFrame(cur, k1["angle"], 640.0, 480.0, 40.0, last, 15.0)
    print("call %.0f us, matches %d" % ((time.perf_counter() - t0) * 1e6, n))
