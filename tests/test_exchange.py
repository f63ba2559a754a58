"""The multi-GPU exchange behind the C ABI (include/plvs_hip.h: plvs_hip_tsdf_exchange_block_lists,
plvs_block_directory): the global block directory built from gathered lists.  The collective itself needs
one process per GPU (the driver's multi-GPU bench runs it); here the lists are fabricated the way
ncclAllGather lays them out, and the owner function is checked against plvs_amd/shard.py."""
import ctypes

import numpy as np
import pytest


@pytest.mark.gpu
def test_block_directory_from_gathered_lists():
    import torch
    from plvs_amd import _lib
    from plvs_amd.shard import owner_of
    lib = _lib.lib
    world, cap = 4, 256
    rng = np.random.default_rng(3)
    ids = np.unique(rng.integers(-40, 40, (600, 3)).astype(np.int32), axis=0)
    own = owner_of(ids, world)
    all_ids = np.zeros((world, cap, 3), np.int32)
    counts = np.zeros(world, np.int32)
    for r in range(world):
        mine = ids[own == r][:cap]
        counts[r] = len(mine)
        all_ids[r, :len(mine)] = mine
    d = ctypes.c_void_p()
    lib.plvs_hip_block_directory_create.argtypes = [ctypes.c_int, ctypes.c_void_p]
    _lib.check(lib.plvs_hip_block_directory_create(4096, ctypes.byref(d)))
    d_ids, d_cnt = torch.from_numpy(all_ids).cuda(), torch.from_numpy(counts).cuda()
    lib.plvs_hip_block_directory_merge.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p]
    for _ in range(2):      # merging the same lists again changes nothing
        _lib.check(lib.plvs_hip_block_directory_merge(d, _lib.t_ptr(d_ids), _lib.t_ptr(d_cnt), world, cap, None))
    n = ctypes.c_int()
    lib.plvs_hip_block_directory_count.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    _lib.check(lib.plvs_hip_block_directory_count(d, ctypes.byref(n)))
    assert n.value == int(counts.sum())
    out_ids = np.zeros((n.value, 3), np.int32)
    out_own = np.zeros(n.value, np.int32)
    lib.plvs_hip_block_directory_list.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    _lib.check(lib.plvs_hip_block_directory_list(d, _lib.np_ptr(out_ids), _lib.np_ptr(out_own), n.value, ctypes.byref(n)))
    got = {tuple(i): int(o) for i, o in zip(out_ids, out_own)}
    want = {tuple(all_ids[r, k]): r for r in range(world) for k in range(counts[r])}
    assert got == want
    assert all(owner_of(np.array([k]), world)[0] == v for k, v in got.items())
    lib.plvs_hip_block_directory_destroy.argtypes = [ctypes.c_void_p]
    lib.plvs_hip_block_directory_destroy(d)
    # the collective refuses a null communicator instead of crashing
    f = lib.plvs_hip_tsdf_exchange_block_lists
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert f(None, _lib.t_ptr(d_ids), 1, cap, _lib.t_ptr(d_ids), _lib.t_ptr(d_cnt), None) == _lib.PLVS_ERR_INVALID_ARG
