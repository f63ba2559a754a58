"""The scenario of tests/test_tsdf_collect.py, run in a process of its own (PLVS_TSDF_COLLECT is read once per process):
an order-free chisel map takes depth-image calls of several lengths and point-stream calls of several clouds (tiles that
at 2 cm reach walk_tiles: the collected chain must hand those calls to the general one); after every call one line
`<visits> <voxels> <sha256 of the whole map>` goes to stdout."""
import hashlib
import sys

import numpy as np


def main():
    import torch
    from plvs_amd.tsdf import TsdfChisel
    from tests import oracle_lib
    from tests.plvs_amd_synth import TUM1, make_rgbd_frames
    from tests.test_tsdf_chisel_depth import _clouds, _integrate_clouds, _integrate_depth

    oracle = oracle_lib.load()
    w, h, step = 640, 480, 2
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    frames = make_rgbd_frames(24, seed=9, holes=True)
    # 5 cm: depth calls of 6, 1 and 8 images, point streams of 3 and 2 clouds in one call, depth again.
    # 2 cm: a tile of a far surface holds more voxels than the largest lean table, some tiles reach walk_tiles.
    plans = [(0.05, 8192, [("depth", 6), ("depth", 1), ("clouds", 3), ("depth", 8), ("clouds", 2), ("depth", 4)]),
             (0.02, 65536, [("depth", 3), ("clouds", 3), ("depth", 2)])]
    for res, max_chunks, plan in plans:
        dev = TsdfChisel(res, max_chunks=max_chunks, order_free=True)
        k0 = 0
        for kind, nb in plan:
            fr, kf = frames[k0:k0 + nb], [300 + k0 + i for i in range(nb)]
            k0 += nb
            if kind == "depth":
                _integrate_depth(dev, fr, grid, step, 0.1, 5.0, kf)
            else:
                _integrate_clouds(dev, _clouds(oracle, fr, grid, step, 0.1, 5.0, kf))
            torch.cuda.synchronize()
            st = dev.last_stats()
            hsh = hashlib.sha256()
            for cid in sorted(tuple(int(v) for v in c) for c in dev.chunk_ids()):
                hsh.update(np.asarray(cid, np.int32).tobytes())
                for plane in dev.get_chunk(*cid):
                    hsh.update(np.ascontiguousarray(plane).tobytes())
            print(st["visits"], st["voxels"], hsh.hexdigest(), flush=True)
        dev.close()


if __name__ == "__main__":
    sys.exit(main())
