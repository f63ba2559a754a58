"""voxblox's "fast" integration method (FastTsdfIntegrator, tsdf_integrator.cc:505-605 — PLVS's YAML default), one thread:
a ray per start voxel of half the voxel size, cast from the surface end, stopped at the third already-seen voxel in a
row, both tests through the reference's lossy ApproxHashSet.  The oracle's sequential loop is pinned by the compiled
reference (tests/test_oracle_pinned.py); here: its properties, the round-based procedure the device uses against that
loop (CPU), and the HIP path against the oracle bit for bit through the C ABI (GPU)."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def rgba_of(kf):
    return np.concatenate([kf["rgb"], np.full((len(kf["rgb"]), 1), 255, np.uint8)], 1)


def maps_equal(ref, hip):
    ids = sorted(tuple(int(v) for v in b) for b in ref.chunk_ids())
    assert ids == sorted(tuple(int(v) for v in b) for b in hip.chunk_ids())
    for bid in ids:
        for name, x, y in zip(("distance", "weight", "colour"), ref.get_chunk(*bid), hip.get_chunk(*bid)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (name, bid)
    return ids


def test_oracle_fast_properties(oracle):
    vs = 0.05
    kfs = make_keyframes(2, seed=2)
    f, s = oracle.voxblox(vs), oracle.voxblox(vs)
    for k in kfs:
        f.integrate_fast(k["xyz"], rgba_of(k), k["Twc"], approx_sets=True)
        s.integrate(k["xyz"], rgba_of(k), k["Twc"])
    # what the integrator is for: a small fraction of simple's voxel updates ...
    assert 0.02 * s.last_visits() < f.last_visits() < 0.25 * s.last_visits()
    # ... on (nearly) the same blocks, every updated voxel also one simple updates, with less weight
    a = {tuple(b) for b in f.chunk_ids()}
    b = {tuple(b) for b in s.chunk_ids()}
    assert a <= b and len(a) > 0.8 * len(b)
    wf = ws = 0.0
    for bid in a:
        df, wgt_f, _ = f.get_chunk(*bid)
        ds, wgt_s, _ = s.get_chunk(*bid)
        assert not ((wgt_f > 0) & ~(wgt_s > 0)).any()
        wf += float(wgt_f.sum())
        ws += float(wgt_s.sum())
    assert wf < 0.5 * ws
    # a second scan of the same cloud starts from "reset" sets: it casts the same rays again
    g = oracle.voxblox(vs)
    g.integrate_fast(kfs[0]["xyz"], rgba_of(kfs[0]), kfs[0]["Twc"], approx_sets=True)
    v1 = g.last_visits()
    g.integrate_fast(kfs[0]["xyz"], rgba_of(kfs[0]), kfs[0]["Twc"], approx_sets=True)
    assert g.last_visits() == v1
    # the collision-free sets give another, close map (the class statement of DESIGN §4.1)
    e = oracle.voxblox(vs)
    for k in kfs:
        e.integrate_fast(k["xyz"], rgba_of(k), k["Twc"], approx_sets=False)
    assert 0.9 * f.last_visits() < e.last_visits() < 1.1 * f.last_visits()


@pytest.mark.parametrize("vs,carving,window", [(0.05, False, 6), (0.10, True, 6), (0.05, False, 1), (0.05, True, 100000)])
def test_round_procedure_equals_the_sequential_loop(oracle, vs, carving, window):
    """oracle_voxblox_fast_model — the stable sort of the set queries by word + every ray reading off its stop, repeated
    until nothing changes: what the device runs — gives every ray the number of updates the sequential loop gives it."""
    L = oracle.lib
    L.oracle_voxblox_fast_model.restype = ctypes.c_int
    L.oracle_voxblox_fast_model.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p]
    L.oracle_voxblox_fast_record_updates.argtypes = [ctypes.c_void_p]
    k = make_keyframes(1, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=5)[0] if carving else make_keyframes(1, seed=5)[0]
    xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
    n = len(xyz)
    Twc = np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)
    o = oracle.voxblox(vs, carving=carving)
    Q, Lm, upd = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    rounds = L.oracle_voxblox_fast_model(o.h, xyz.ctypes.data, n, Twc.ctypes.data, window, Q.ctypes.data, Lm.ctypes.data)
    L.oracle_voxblox_fast_record_updates(upd.ctypes.data)
    o.integrate_fast(xyz, rgba_of(k)[::2], Twc, approx_sets=True)
    assert np.array_equal(Lm, upd)
    assert 2 <= rounds < 200 and Lm.sum() == o.last_visits() > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("vs,carving,far", [(0.05, False, False), (0.10, True, True), (0.02, False, False), (0.05, True, False)])
def test_hip_fast_matches_oracle(oracle, vs, carving, far):
    from plvs_amd.tsdf import TsdfVoxblox
    kfs = make_keyframes(4, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=21) if far else make_keyframes(4, seed=21)
    ref = oracle.voxblox(vs, carving=carving)
    hip = TsdfVoxblox(vs, use_carving=carving, max_blocks=8192 if vs >= 0.05 else 16384)
    for i, k in enumerate(kfs):
        ref.integrate_fast(k["xyz"], rgba_of(k), k["Twc"], approx_sets=True)
        hip.integrate_fast(k["xyz"], rgba_of(k), k["Twc"])
        assert hip.last_stats()["visits"] == ref.last_visits(), i
        assert 2 <= hip.fast_rounds() < 200
    assert len(maps_equal(ref, hip)) > 20
    hip.close()


@pytest.mark.gpu
def test_hip_fast_batches_scans_history_and_other_methods(oracle):
    """Several clouds in one call are several scans; the sets keep their words from scan to scan (an index whose hash is 0
    depends on it); the methods can alternate on one map; an empty scan still moves the sets on; clear() starts over."""
    import torch
    from plvs_amd.tsdf import TsdfVoxblox
    vs = 0.05
    kfs = make_keyframes(5, seed=33)
    # the world origin's voxel (hash 0) inside the view: a camera 1 m in front of it
    ref, hip = oracle.voxblox(vs), TsdfVoxblox(vs, max_blocks=8192)
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs[:3]])).cuda()
    rgba = torch.from_numpy(np.concatenate([rgba_of(k) for k in kfs[:3]])).cuda()
    Twc = torch.from_numpy(np.stack([np.asarray(k["Twc"], np.float32).reshape(3, 4) for k in kfs[:3]])).cuda()
    offsets = np.cumsum([0] + [len(k["xyz"]) for k in kfs[:3]]).astype(np.int32)
    hip.integrate_fast_batch_dev(xyz, rgba, offsets, Twc)
    torch.cuda.synchronize()
    for k in kfs[:3]:
        ref.integrate_fast(k["xyz"], rgba_of(k), k["Twc"], approx_sets=True)
    maps_equal(ref, hip)
    # simple, an empty scan, fast again
    ref.integrate(kfs[3]["xyz"], rgba_of(kfs[3]), kfs[3]["Twc"])
    hip.integrate(kfs[3]["xyz"], rgba_of(kfs[3]), kfs[3]["Twc"])
    ref.integrate_fast(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[3]["Twc"], approx_sets=True)
    hip.integrate_fast(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[3]["Twc"])
    ref.integrate_fast(kfs[4]["xyz"], rgba_of(kfs[4]), kfs[4]["Twc"], approx_sets=True)
    hip.integrate_fast(kfs[4]["xyz"], rgba_of(kfs[4]), kfs[4]["Twc"])
    maps_equal(ref, hip)
    # points around the world origin (voxel (0, 0, 0): hash 0) seen from 1 m away, twice
    g = np.stack(np.meshgrid(np.linspace(-0.2, 0.2, 41), np.linspace(-0.2, 0.2, 41), [1.0]), -1).reshape(-1, 3).astype(np.float32)
    T = np.array([[1, 0, 0, 0.01], [0, 1, 0, 0.01], [0, 0, 1, -0.98]], np.float32)
    col = np.full((len(g), 4), 200, np.uint8)
    for _ in range(2):
        ref.integrate_fast(g, col, T, approx_sets=True)
        hip.integrate_fast(g, col, T)
    maps_equal(ref, hip)
    hip.clear()
    ref2 = oracle.voxblox(vs)
    ref2.integrate_fast(kfs[0]["xyz"], rgba_of(kfs[0]), kfs[0]["Twc"], approx_sets=True)
    hip.integrate_fast(kfs[0]["xyz"], rgba_of(kfs[0]), kfs[0]["Twc"])
    maps_equal(ref2, hip)
    hip.close()


@pytest.mark.gpu
def test_hip_fast_on_a_sharded_map_is_the_union_of_its_shards(oracle):
    """Block-hash shards (the multi-GPU layout): every rank settles the same rays (the sets do not know about owners) and
    keeps the voxels of its own blocks — the shards' blocks are disjoint and together they are the one-GPU map."""
    from plvs_amd.tsdf import TsdfVoxblox
    vs = 0.05
    kfs = make_keyframes(3, seed=44)
    whole = TsdfVoxblox(vs, max_blocks=8192)
    shards = [TsdfVoxblox(vs, max_blocks=8192, shard_rank=r, shard_count=3) for r in range(3)]
    for k in kfs:
        whole.integrate_fast(k["xyz"], rgba_of(k), k["Twc"])
        for sh in shards:
            sh.integrate_fast(k["xyz"], rgba_of(k), k["Twc"])
    ids = sorted(tuple(int(v) for v in b) for b in whole.chunk_ids())
    seen = {}
    for r, sh in enumerate(shards):
        for b in sh.chunk_ids():
            bid = tuple(int(v) for v in b)
            assert bid not in seen
            seen[bid] = r
    assert sorted(seen) == ids and len(set(seen.values())) == 3
    for bid in ids:
        for x, y in zip(whole.get_chunk(*bid), shards[seen[bid]].get_chunk(*bid)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), bid
    for t in shards + [whole]:
        t.close()


_SEQUENTIAL_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes
from tests.test_tsdf_voxblox_fast import maps_equal, rgba_of
from plvs_amd.tsdf import TsdfVoxblox
oracle = oracle_lib.load()
kfs = make_keyframes(3, seed=5)
ref, hip = oracle.voxblox(0.05, carving=True), TsdfVoxblox(0.05, use_carving=True, max_blocks=8192)
for k in kfs:
    xyz, col = np.ascontiguousarray(k["xyz"][::6]), np.ascontiguousarray(rgba_of(k)[::6])
    ref.integrate_fast(xyz, col, k["Twc"], approx_sets=True)
    hip.integrate_fast(xyz, col, k["Twc"])
    assert hip.last_stats()["visits"] == ref.last_visits() > 1000
    assert hip.fast_rounds() <= int(sys.argv[2])
print("blocks", len(maps_equal(ref, hip)))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("max_rounds", [0, 2], ids=["one_thread_from_the_start", "after_two_rounds"])
def test_hip_fast_finishes_on_one_thread_when_the_rounds_do_not_settle(max_rounds, tmp_path):
    """the rounds of vb_fast_plan can take one iteration per ray on an adversarial cloud; past
    PLVS_VB_FAST_MAX_ROUNDS (512) the plan is finished by vbf_sequential — one device thread, the reference's own order.
    Forced here to 0 and 2 rounds (the switch is read once per process: a child process): the maps equal the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "seq.py"
    script.write_text(_SEQUENTIAL_SCRIPT)
    env = dict(os.environ, PLVS_VB_FAST_MAX_ROUNDS=str(max_rounds))
    r = subprocess.run([sys.executable, str(script), root, str(max_rounds)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert int(r.stdout.strip().split()[-1]) > 10
