"""walk_multi (round 6, plvs_amd/csrc/tsdf_walk_multi.hpp): the order-free walk of a long depth-image call takes the same
block of grid pixels of several consecutive key frames into one voxel table, and the first part of the call tells the rest
which voxels' colours have saturated.

Bar (GPU): whatever the grouping — one image per workgroup (round 5's walk_fast), 2 / 3 / 4 / 7 images per task, with and
without the saturation feedback — the maps are BIT-IDENTICAL: sdf, weight, kfid, colour and colour weight of every voxel of
every chunk, and the visit counts of every call.  (The grouping changes which records and colour runs leave the walk, not
what apply_chunks sums or what the fold folds: Chisel.cpp:505-540, ColorVoxel.h:91-110.)  Against the oracle: the depth
entry point's own tests (test_tsdf_chisel_depth.py, test_measured_configs.py) run with the library's default grouping."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_rgbd_frames
from tests.test_tsdf_chisel import compare_maps
from tests.test_tsdf_chisel_depth import _clouds, _integrate_clouds, _integrate_depth


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def _dense_frames(n, seed=0, first=0):
    """Key frames of a slowly moving camera (6 mm, 0.15 deg per frame: consecutive frames see nearly the same voxels) in
    the 6 x 4 x 3 m room of make_rgbd_frames, with holes; every 7th frame jumps (a window that must close early)."""
    from tests.plvs_amd_synth import _m
    rng = np.random.default_rng(seed)
    half = np.array((6.0, 4.0, 3.0)) / 2.0
    spheres = [(np.array([2.3, 0.5, 0.2]), 0.4), (np.array([-2.3, -0.4, -0.3]), 0.4), (np.array([-2.0, 1.3, 0.4]), 0.4)]
    out = []
    for k in range(first, first + n):
        yaw = np.deg2rad(0.15 * k + (25.0 if (k % 7) == 6 else 0.0))
        pos = np.array([0.3 + 0.006 * k, -0.2, 0.05 * np.sin(0.1 * k)])
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        R = np.stack([right, np.cross(fwd, right), fwd], axis=1)
        _, _, depth = _m._render_depth(R, pos, TUM1, (-half, half), spheres, 1)
        depth = depth + rng.standard_normal(depth.shape) * (0.0012 + 0.0019 * (depth - 0.4) ** 2)
        d32 = depth.astype(np.float32)
        h, w = d32.shape
        for _ in range(6):
            y0, x0 = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 40))
            d32[y0:y0 + int(rng.integers(3, 40)), x0:x0 + int(rng.integers(3, 40))] = (0.0, np.nan, 25.0)[int(rng.integers(0, 3))]
        bgr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        out.append(dict(depth=np.ascontiguousarray(d32), bgr=bgr,
                        Twc=np.ascontiguousarray(np.concatenate([R, pos[:, None]], axis=1).astype(np.float32))))
    return out


GROUPINGS = [(1, 1), (4, 4), (2, 2), (7, 1), (3, 8), (0, 0)]   # (images per task, first-part divisor); (1, 1) = walk_fast


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["dense_stream", "turning_camera"])
def test_hip_walk_windows_maps_do_not_depend_on_the_grouping(scene):
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = 640, 480, 2
    oracle = oracle_lib.load()
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    calls = (16, 9, 23)
    total = sum(calls)
    frames = _dense_frames(total, seed=3) if scene == "dense_stream" else make_rgbd_frames(total, seed=7, holes=True)
    maps = []
    for per_task, part in GROUPINGS:
        t = TsdfChisel(0.05, max_chunks=8192, order_free=True)
        t.set_walk_windows(per_task, part)
        maps.append(t)
    k0 = 0
    for nb in calls:
        fr, kf = frames[k0:k0 + nb], [500 + k0 + i for i in range(nb)]
        k0 += nb
        visits = []
        for t in maps:
            _integrate_depth(t, fr, grid, step, 0.1, 5.0, kf)
            visits.append(t.last_stats()["visits"])
        assert len(set(visits)) == 1 and visits[0] > 0, visits
        for t in maps[1:]:          # after EVERY call: colour weights below 254 in the first ones
            assert compare_maps(maps[0], t, tol=0.0) >= 2
    for t in maps:
        t.close()


@pytest.mark.gpu
def test_hip_walk_windows_against_the_point_stream_entry(oracle):
    """... and against the oracle's clouds through the point-stream entry point (round 4's kernels, 512 consecutive points per
    tile): the same integers, the same colours, call by call."""
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = 640, 480, 2
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    frames = _dense_frames(20, seed=9)
    a = TsdfChisel(0.05, max_chunks=8192, order_free=True)
    b = TsdfChisel(0.05, max_chunks=8192, order_free=True)
    for k0, nb in ((0, 12), (12, 8)):
        fr, kf = frames[k0:k0 + nb], [7 + k0 + i for i in range(nb)]
        _integrate_depth(a, fr, grid, step, 0.1, 5.0, kf)
        _integrate_clouds(b, _clouds(oracle, fr, grid, step, 0.1, 5.0, kf))
        assert a.last_stats()["visits"] == b.last_stats()["visits"] > 0
        assert compare_maps(a, b, tol=0.0) >= 2
    a.close()
    b.close()


@pytest.mark.gpu
def test_hip_walk_windows_voxel_with_more_visits_than_a_record_holds():
    """A wall 30 cm in front of the camera at 20 cm voxels: every ray of a block visits the same few voxels, 512 per image —
    a window closes before the 32-bit sums of a record could overflow (kMultiCountCap), whatever the task length."""
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = 640, 480, 2
    oracle = oracle_lib.load()
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    rng = np.random.default_rng(5)
    frames = []
    for k in range(9):
        d = (0.30 + 0.01 * rng.random((h, w))).astype(np.float32)
        Twc = np.eye(4, dtype=np.float32)[:3].copy()
        Twc[0, 3] = 0.002 * k
        frames.append(dict(depth=d, bgr=rng.integers(0, 256, (h, w, 3), dtype=np.uint8), Twc=Twc))
    a = TsdfChisel(0.20, max_chunks=1024, order_free=True)
    b = TsdfChisel(0.20, max_chunks=1024, order_free=True)
    a.set_walk_windows(1, 1)
    b.set_walk_windows(7, 3)
    for t in (a, b):
        _integrate_depth(t, frames, grid, step, 0.1, 5.0, list(range(9)))
    assert a.last_stats() == b.last_stats() and a.last_stats()["max_run"] > 4 * 512
    assert compare_maps(a, b, tol=0.0) >= 1
    a.close()
    b.close()
