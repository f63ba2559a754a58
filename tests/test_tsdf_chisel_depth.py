"""Depth images straight into the chisel map (plvs_hip_tsdf_chisel_integrate_depth_batch_dev, round 5):
GeneratePointCloudInCameraFrameBGRA (src/PointCloudMapping.cc:957-996) + InsertCloud in one call, the cloud never written.

Bar (GPU): the map of the depth entry point is BIT-IDENTICAL — sdf, weight, kfid, colour, colour weight of every voxel of
every chunk — to the map of the cloud the ORACLE's generator makes from the same images (oracle/cloudgen.c) integrated through
the point-stream entry point of the same mode; and, in the ordered mode, to the oracle's own sequential integrate of that
cloud.  The order-free walk of the depth entry point uses 32 x 16 blocks of grid pixels as tiles where the point-stream
walk uses 512 consecutive points: equality of the two maps is equality of the integer sums of the same visits, of the
last visitors in the reference's point order and of the colour fold's visit order across tiles."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_rgbd_frames
from tests.test_tsdf_chisel import compare_maps


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def _clouds(oracle, frames, grid, step, min_depth, max_depth, kfids):
    out = []
    for fr, k in zip(frames, kfids):
        rec, _ = oracle.cloudgen(fr["depth"], fr["bgr"], grid, step, min_depth, max_depth, int(k))
        out.append(dict(xyz=np.stack([rec["x"], rec["y"], rec["z"]], -1).astype(np.float32),
                        rgb=np.stack([rec["r"], rec["g"], rec["b"]], -1).astype(np.uint8),
                        kfid=rec["kfid"].astype(np.uint32), Twc=fr["Twc"]))
    return out


def _integrate_clouds(dev, clouds):
    import torch
    xyz = torch.from_numpy(np.concatenate([c["xyz"] for c in clouds])).cuda()
    rgb = torch.from_numpy(np.concatenate([c["rgb"] for c in clouds])).cuda()
    kfid = torch.from_numpy(np.concatenate([c["kfid"] for c in clouds]).astype(np.int32)).cuda()
    off = np.cumsum([0] + [c["xyz"].shape[0] for c in clouds]).astype(np.int32)
    Twc = torch.from_numpy(np.stack([c["Twc"] for c in clouds])).cuda()
    dev.integrate_batch_dev(xyz, rgb, kfid, off, Twc)


def _integrate_depth(dev, frames, grid, step, min_depth, max_depth, kfids, pitched=False):
    import torch
    depth = torch.from_numpy(np.stack([f["depth"] for f in frames])).cuda()
    bgr = torch.from_numpy(np.stack([f["bgr"] for f in frames])).cuda()
    if pitched:   # rows and images further apart than they need to be
        n, h, w = depth.shape
        big = torch.full((n, h + 3, w + 5), float("nan"), dtype=torch.float32, device="cuda")
        big[:, :h, :w] = depth
        depth = big[:, :h, :w]
        bigc = torch.zeros((n, h + 2, w + 7, 3), dtype=torch.uint8, device="cuda")
        bigc[:, :h, :w] = bgr
        bgr = bigc[:, :h, :w]
    dev.integrate_depth_batch_dev(depth, bgr, torch.from_numpy(grid).cuda(), step, min_depth, max_depth,
                                  torch.from_numpy(np.asarray(kfids, np.int32)).cuda(),
                                  torch.from_numpy(np.stack([f["Twc"] for f in frames])).cuda())


def _crop(frames, width, height):
    return [dict(depth=np.ascontiguousarray(f["depth"][:height, :width]), bgr=np.ascontiguousarray(f["bgr"][:height, :width]),
                 Twc=f["Twc"]) for f in frames]


CASES = [
    dict(id="tum_step2_holes", width=640, height=480, step=2, res=0.05, batches=(3, 2, 1)),
    dict(id="full_res_step1", width=640, height=480, step=1, res=0.05, batches=(2,)),
    dict(id="odd_step3_pitched", width=637, height=479, step=3, res=0.10, batches=(2, 2), pitched=True),
    dict(id="small_image", width=100, height=70, step=1, res=0.05, batches=(1, 2)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("order_free", [True, False], ids=["order_free", "ordered"])
@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_hip_depth_entry_equals_cloudgen_plus_integrate(oracle, case, order_free):
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = case["width"], case["height"], case["step"]
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    total = sum(case["batches"])
    frames = _crop(make_rgbd_frames(total, seed=5, holes=True), w, h)
    a = TsdfChisel(case["res"], max_chunks=8192, order_free=order_free)     # depth entry point
    b = TsdfChisel(case["res"], max_chunks=8192, order_free=order_free)     # oracle's clouds through the point-stream entry
    ora = oracle.chisel(case["res"]) if not order_free else None
    k0 = 0
    for nb in case["batches"]:
        fr, kf = frames[k0:k0 + nb], [100 + k0 + i for i in range(nb)]
        k0 += nb
        _integrate_depth(a, fr, grid, step, 0.1, 5.0, kf, pitched=case.get("pitched", False))
        clouds = _clouds(oracle, fr, grid, step, 0.1, 5.0, kf)
        _integrate_clouds(b, clouds)
        assert a.last_stats()["visits"] == b.last_stats()["visits"] > 0
        if ora is not None:
            for c in clouds:
                ora.integrate(c["xyz"], c["rgb"], c["kfid"], c["Twc"])
        n = compare_maps(a, b)     # after EVERY call: colour weights below 254 everywhere in the first ones
        assert n >= 2
        if ora is not None:
            compare_maps(ora, a)
    a.close()
    b.close()


@pytest.mark.gpu
def test_hip_depth_entry_voxel_seen_by_many_tiles_of_a_band(oracle):
    """A wall 15 cm in front of the camera at 25 cm voxels: one voxel is visited by rays of more than eight column blocks
    of a band of grid rows — the general form of the fold's band merge — and by most bands of the image."""
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = 640, 480, 2
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    rng = np.random.default_rng(11)
    frames = []
    for k in range(3):
        d = (0.15 + 0.02 * rng.random((h, w))).astype(np.float32)
        d[rng.random((h, w)) < 0.05] = 0.0
        Twc = np.eye(4, dtype=np.float32)[:3].copy()
        Twc[0, 3] = 0.01 * k
        frames.append(dict(depth=d, bgr=rng.integers(0, 256, (h, w, 3), dtype=np.uint8), Twc=Twc))
    for order_free in (True, False):
        a = TsdfChisel(0.25, max_chunks=1024, order_free=order_free)
        b = TsdfChisel(0.25, max_chunks=1024, order_free=order_free)
        for k, fr in enumerate(frames):
            _integrate_depth(a, [fr], grid, step, 0.1, 5.0, [k])
            _integrate_clouds(b, _clouds(oracle, [fr], grid, step, 0.1, 5.0, [k]))
            compare_maps(a, b)
        assert a.last_stats()["max_run"] > 8 * 32 * 2 or not order_free
        a.close()
        b.close()


@pytest.mark.gpu
def test_hip_depth_entry_empty_and_invalid_images(oracle):
    from plvs_amd.tsdf import TsdfChisel
    w, h, step = 320, 240, 2
    grid = oracle.cam_grid_points(w, h, step, TUM1["fx"] / 2, TUM1["fy"] / 2, TUM1["cx"] / 2, TUM1["cy"] / 2)
    Twc = np.eye(4, dtype=np.float32)[:3]
    for order_free in (True, False):
        a = TsdfChisel(0.05, max_chunks=1024, order_free=order_free)
        for fill in (0.0, np.nan, 25.0, -1.0):      # no return, NaN, beyond max_depth, negative: no point at all
            fr = dict(depth=np.full((h, w), fill, np.float32), bgr=np.zeros((h, w, 3), np.uint8), Twc=Twc)
            _integrate_depth(a, [fr, fr], grid, step, 0.1, 5.0, [1, 2])
            assert a.num_chunks() == 0 and a.last_stats()["visits"] == 0
        # one valid pixel in the last grid row / column
        d = np.zeros((h, w), np.float32)
        d[h - 2, w - 2] = 1.0
        fr = dict(depth=d, bgr=np.full((h, w, 3), 200, np.uint8), Twc=Twc)
        _integrate_depth(a, [fr], grid, step, 0.1, 5.0, [9])
        b = TsdfChisel(0.05, max_chunks=1024, order_free=order_free)
        _integrate_clouds(b, _clouds(oracle, [fr], grid, step, 0.1, 5.0, [9]))
        assert compare_maps(a, b) >= 1 and a.last_stats()["visits"] == b.last_stats()["visits"] > 0
        a.close()
        b.close()


_SPIN_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_rgbd_frames
from tests.test_tsdf_chisel import compare_maps
from tests.test_tsdf_chisel_depth import _clouds, _integrate_clouds, _integrate_depth
from plvs_amd.tsdf import TsdfChisel
oracle = oracle_lib.load()
grid = oracle.cam_grid_points(640, 480, 2, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
frames = make_rgbd_frames(4, seed=9, holes=True)
a, b = TsdfChisel(0.05, max_chunks=8192, order_free=True), TsdfChisel(0.05, max_chunks=8192, order_free=True)
for k0 in (0, 2):
    fr, kf = frames[k0:k0 + 2], [k0, k0 + 1]
    _integrate_depth(a, fr, grid, 2, 0.1, 5.0, kf)
    _integrate_clouds(b, _clouds(oracle, fr, grid, 2, 0.1, 5.0, kf))
    assert a.last_stats()["visits"] == b.last_stats()["visits"] > 0
    print("chunks", compare_maps(a, b))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("spin_us", ["0", "1"], ids=["sleeps_on_the_stream", "gives_up_polling_after_a_microsecond"])
def test_hip_counter_read_without_polling(spin_us, tmp_path):
    """The host normally POLLS a word in pinned memory for the call's counters (wait_published, PLVS_TSDF_SPIN_US = 3000); with
    0 it sleeps on the stream as it did before, with 1 it starts polling and falls back to the stream at once — the two
    maps of the depth / point-stream entry points stay bit-identical either way (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "spin.py"
    script.write_text(_SPIN_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, PLVS_TSDF_SPIN_US=spin_us), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("chunks") == 2
