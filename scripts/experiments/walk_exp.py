"""Order-free integrate on the bench stream with parts of walk_acc switched off (PLVS_WALK_EXP bits:
1 = no accumulator atomics, 2 = no table, 4 = no flush).  Prints the stage times."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402

kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
for exp in [int(a) for a in sys.argv[1:]] or [0]:
    pass
    t = TsdfChisel(0.05, max_chunks=16384, order_free=exp >= 0)   # a negative argument: the ordered mode
    for _ in range(2):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    t.set_profiling(True)
    for _ in range(4):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    sm, c = t.stage_ms()
    print(exp, {k: round(v / c, 4) for k, v in sm.items()}, t.last_stats())
    t.close()
