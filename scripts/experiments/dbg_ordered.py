import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import oracle_lib
from plvs_amd.synth_scene import make_keyframes, TUM1
from plvs_amd.tsdf import TsdfChisel
oracle = oracle_lib.load()
def small_cam(scale):
    c = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        c[k] = c[k] / scale
    c["width"] //= scale
    c["height"] //= scale
    return c
for rep in range(6):
  for res, nkf, scale in [(0.10, 3, 2)]:
   print("case", res, nkf, scale)
   ora = oracle.chisel(res)
   dev = TsdfChisel(res, max_chunks=4096, order_free=bool(int(os.environ.get('ORDER_FREE', '0'))))
   for it, kf in enumerate(make_keyframes(nkf, cam=small_cam(scale), seed=11)):
       ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
       dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
       print("kf", it, dev.last_stats(), ora.last_visits())
       ia = {tuple(x) for x in ora.chunk_ids()}
       ib = {tuple(x) for x in dev.chunk_ids()}
       print(" chunks", len(ia), len(ib), ia == ib, "dev list", len(dev.chunk_ids()))
       for cid in sorted(ia & ib):
           a, b = ora.get_chunk(*cid), dev.get_chunk(*cid)
           if int(os.environ.get('ORDER_FREE', '0')):
               ds = np.flatnonzero(np.abs(a[0] - b[0]) > 2e-5)
               dw = np.flatnonzero(np.abs(a[1] - b[1]) > 5e-5 * np.maximum(a[1], 1e-9))
           else:
               ds = np.flatnonzero(a[0].view(np.uint32) != b[0].view(np.uint32))
               dw = np.flatnonzero(a[1].view(np.uint32) != b[1].view(np.uint32))
           if len(ds) or len(dw):
               print("  chunk", cid, "sdf diffs", len(ds), "w diffs", len(dw), "known ora", int((a[1] > 0).sum()), "known dev", int((b[1] > 0).sum()))
               for v in ds[:4]:
                   print("    vox", v, "ora", a[0][v], a[1][v], "dev", b[0][v], b[1][v])
   dev.close()

