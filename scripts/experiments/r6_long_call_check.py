import hashlib, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from plvs_amd.tsdf import TsdfChisel
from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_rgbd_frames
from tests.test_tsdf_chisel_depth import _integrate_depth
oracle = oracle_lib.load()
w, h, step = 640, 480, 2
grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
frames = make_rgbd_frames(24, seed=3, holes=True)
dev = TsdfChisel(0.05, max_chunks=8192, order_free=True)
for rep in range(3):
    fr = frames * 10            # 240 images in ONE call: 36 000 tiles
    _integrate_depth(dev, fr, grid, step, 0.1, 5.0, list(range(rep * 1000, rep * 1000 + len(fr))))
    torch.cuda.synchronize()
    st = dev.last_stats()
    hsh = hashlib.sha256()
    for cid in sorted(tuple(int(v) for v in c) for c in dev.chunk_ids()):
        for plane in dev.get_chunk(*cid):
            hsh.update(np.ascontiguousarray(plane).tobytes())
    print(st["visits"], st["voxels"], hsh.hexdigest()[:16], flush=True)
