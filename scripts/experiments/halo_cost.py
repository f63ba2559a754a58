#!/usr/bin/env python3
"""Cost of meshing a ray-sharded chisel map, emulated on ONE device with virtual ranks: after a 100-keyframe
batch (the bench's scene) every rank meshes its chunks of meshesToUpdate; the halo (chunks of other ranks the
meshes read) is moved with tensor copies (not timed as communication).  Printed per N: chunks meshed per rank,
chunks fetched per rank and their size, rounds, and the slowest rank's wall time of probe passes, export + import
kernels and the final mesh call — beside the single-device mesh_chunks of the same list."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from plvs_amd.shard import owner_of  # noqa: E402
from tests.synth_scene import make_keyframes  # noqa: E402
from plvs_amd.tsdf import TsdfChisel  # noqa: E402
from tests.test_shard_rays import _batch, _nbhd27, sharded_step, virtual_halo_round  # noqa: E402

WORLDS = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz, rgb, kfid, offsets, Twc = _batch(kfs)


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t) * 1e3


single = TsdfChisel(0.05, max_chunks=16384, order_free=True)
single.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
todo = np.array(_nbhd27(single.updated_chunk_ids()), np.int32)
single.mesh_chunks(todo)
m, ms1 = wall(lambda: single.mesh_chunks(todo))
print("single device: %d chunks in meshesToUpdate (%d exist), %d vertices, mesh_chunks %.2f ms (with the copy to the host)"
      % (len(todo), single.num_chunks(), len(m["vertices"]), ms1))
for world in WORLDS:
    ranks = [TsdfChisel(0.05, max_chunks=16384, shard_rank=r, shard_count=world, order_free=True) for r in range(world)]
    sharded_step(ranks, xyz, rgb, kfid, offsets, Twc)
    own = owner_of(todo, world)
    mine = [np.ascontiguousarray(todo[own == r]) for r in range(world)]
    rounds, fetched, t_probe, t_move, moved = 0, np.zeros(world, np.int64), np.zeros(world), 0.0, 0
    while True:
        missing = []
        for r, t in enumerate(ranks):
            n, ms = wall(lambda: t.mesh_probe(mine[r]))
            t_probe[r] += ms
            missing.append(t.halo_missing() if n else np.zeros((0, 3), np.int32))
        if not any(len(x) for x in missing):
            break
        rounds += 1
        for r in range(world):
            fetched[r] += len(missing[r])
        mv, ms = wall(lambda: virtual_halo_round(ranks, missing))
        moved += mv
        t_move += ms
    t_mesh = [wall(lambda: t.mesh_chunks(mine[r]))[1] for r, t in enumerate(ranks)]
    print("N=%d: %d own chunks to mesh per rank (max), %d asked for per rank (max), %d rounds; slowest rank: probes %.2f ms,"
          " final mesh %.2f ms; export+import kernels of all ranks %.2f ms" %
          (world, max(len(x) for x in mine), fetched.max(), rounds, t_probe.max(), max(t_mesh), t_move) + "; %d chunks moved in all (%.1f MB)" % (moved, moved * 65536 / 1e6))
    for t in ranks:
        t.close()
