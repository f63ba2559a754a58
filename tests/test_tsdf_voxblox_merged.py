"""voxblox's "merged" integration method (MergedTsdfIntegrator, tsdf_integrator.cc:329-492, one thread): the points that
end in one voxel are folded into one ray; the bundles are integrated in the iteration order of the reference's hash
map.  The oracle's properties on CPU; the HIP path against the oracle bit for bit through the C ABI."""
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


def test_oracle_merged_properties(oracle):
    vs = 0.05
    k = make_keyframes(1, seed=2)[0]
    m = oracle.voxblox(vs)
    nb, firsts = m.integrate_merged(k["xyz"], rgba_of(k), k["Twc"])
    n = len(k["xyz"])
    # a 76 800-point keyframe on 5 cm voxels: several points per end voxel -> far fewer rays than points
    assert 0.02 * n < nb < 0.7 * n
    assert len(np.unique(firsts)) == nb and firsts.min() >= 0 and firsts.max() < n
    # the bundles do NOT come in point order (hash-map iteration), nor in the mixed visiting order
    assert not np.array_equal(firsts, np.sort(firsts))
    s = oracle.voxblox(vs)
    s.integrate(k["xyz"], rgba_of(k), k["Twc"])
    assert m.last_visits() < 0.8 * s.last_visits()           # fewer rays, fewer voxel updates
    # same surface: the blocks the merged rays touch are among those of the per-point rays, and nearly all of them
    a = {tuple(b) for b in m.chunk_ids()}
    b = {tuple(b) for b in s.chunk_ids()}
    assert a <= b and len(a) > 0.9 * len(b)
    # with carving, points beyond max_ray_length become clearing bundles (first point only), integrated after the others
    kf = make_keyframes(1, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=3)[0]
    c = oracle.voxblox(vs, carving=True)
    nbc, _ = c.integrate_merged(kf["xyz"], rgba_of(kf), kf["Twc"])
    far = (np.linalg.norm(kf["xyz"], axis=1) > 5.0).sum()
    assert far > 1000 and nbc > 0
    nc = oracle.voxblox(vs, carving=False)
    nbn, _ = nc.integrate_merged(kf["xyz"], rgba_of(kf), kf["Twc"])
    assert nbn < nbc                                          # without carving the far points are dropped


@pytest.mark.gpu
@pytest.mark.parametrize("vs,carving,far", [(0.05, False, False), (0.10, True, True), (0.02, False, False)])
def test_hip_merged_matches_oracle(oracle, vs, carving, far):
    from plvs_amd.tsdf import TsdfVoxblox
    ref, hip = oracle.voxblox(vs, carving=carving), TsdfVoxblox(vs, use_carving=carving, max_blocks=65536)
    kfs = make_keyframes(3, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=5) if far else make_keyframes(3, seed=5)
    if vs < 0.05:
        kfs = [dict(k, xyz=k["xyz"][::3], rgb=k["rgb"][::3]) for k in kfs]
    for i, k in enumerate(kfs):
        if i == 1:                                            # methods may alternate on one map
            ref.integrate(k["xyz"], rgba_of(k), k["Twc"])
            hip.integrate(k["xyz"], rgba_of(k), k["Twc"])
        else:
            ref.integrate_merged(k["xyz"], rgba_of(k), k["Twc"])
            hip.integrate_merged(k["xyz"], rgba_of(k), k["Twc"])
        assert hip.last_stats()["visits"] == ref.last_visits(), i
        assert len(hip.updated_chunk_ids()) > 0
    assert len(maps_equal(ref, hip)) > 10
    hip.integrate_merged(np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), kfs[0]["Twc"])
    assert hip.last_stats()["visits"] == 0
    # a cloud with nothing valid (all closer than min_ray_length) integrates nothing
    near = np.full((10, 3), 0.01, np.float32)
    hip.integrate_merged(near, np.zeros((10, 4), np.uint8), kfs[0]["Twc"])
    assert hip.last_stats()["visits"] == 0
    hip.close()


@pytest.mark.gpu
def test_mirror_runs_the_merged_method(oracle):
    from plvs_amd.tsdf import PointCloudMapVoxblox

    class Merged(PointCloudMapVoxblox):
        skIntegrationMethod = "merged"

    class Fast(PointCloudMapVoxblox):
        skIntegrationMethod = "fast"

    class Unknown(PointCloudMapVoxblox):
        skIntegrationMethod = "quick"

    with pytest.raises(ValueError):
        Unknown(0.05)
    # "fast" (the reference's YAML default): FastTsdfIntegrator's one-thread schedule with its approximate sets
    pf, refs = Fast(0.05), oracle.voxblox(0.05)
    for k in make_keyframes(2, seed=8):
        pf.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
        refs.integrate_fast(k["xyz"], rgba_of(k), k["Twc"], approx_sets=True)
    maps_equal(refs, pf.tsdf)
    pm, ref = Merged(0.05), oracle.voxblox(0.05)
    for k in make_keyframes(2, seed=8):
        pm.InsertCloud(dict(xyz=k["xyz"], rgba=rgba_of(k)), k["Twc"])
        ref.integrate_merged(k["xyz"], rgba_of(k), k["Twc"])
    maps_equal(ref, pm.tsdf)
    assert len(pm.UpdateMap()) > 3000
