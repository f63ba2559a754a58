"""Depth image -> camera-frame cloud: the oracle's properties on CPU, and the HIP
path against the oracle, byte for byte, through the C ABI."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.plvs_amd_synth import TUM1, make_keyframes, make_rgbd_frames


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def _grid(oracle, cam=TUM1, step=2, width=None, height=None):
    return oracle.cam_grid_points(width or cam["width"], height or cam["height"], step, cam["fx"], cam["fy"],
                                  cam["cx"], cam["cy"])


def test_oracle_cloud_matches_the_generator_formula(oracle):
    """Same p = (gx*d, gy*d, d) / min < d < max rule as the numpy generator the TSDF tests feed on."""
    fr = make_rgbd_frames(1, holes=False)[0]
    grid = _grid(oracle)
    rec, p2p = oracle.cloudgen(fr["depth"], fr["bgr"], grid, 2, 0.1, 5.0, 7)
    d = fr["depth"][::2, ::2]
    keep = (d > np.float32(0.1)) & (d < np.float32(5.0))
    assert rec.shape[0] == int(keep.sum()) > 50000
    g = grid.reshape(d.shape[0], d.shape[1], 2)
    np.testing.assert_array_equal(rec["z"], d[keep])
    np.testing.assert_array_equal(rec["x"], (g[..., 0] * d)[keep])
    np.testing.assert_array_equal(rec["y"], (g[..., 1] * d)[keep])
    np.testing.assert_array_equal(rec["depth"], rec["z"])
    assert (rec["kfid"] == 7).all() and (rec["label"] == 0).all() and (rec["a"] == 0).all()
    # r,g,b members take the image's B,G,R bytes (src/PointCloudMapping.cc:978-980)
    c = fr["bgr"][::2, ::2][keep]
    np.testing.assert_array_equal(np.stack([rec["r"], rec["g"], rec["b"]], -1), c)
    # pixelToPointIndex: running index on valid grid pixels, -1 elsewhere
    assert (p2p[1::2] == -1).all() and (p2p[:, 1::2] == -1).all()
    np.testing.assert_array_equal(p2p[::2, ::2][keep], np.arange(rec.shape[0]))
    assert (p2p[::2, ::2][~keep] == -1).all()


def test_oracle_normals(oracle):
    h, w = 48, 64
    cam = dict(fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    grid = oracle.cam_grid_points(w, h, 2, **cam)
    bgr = np.zeros((h, w, 3), np.uint8)
    # fronto-parallel wall: every interior normal is (0, 0, -1) (towards the camera)
    rec, p2p = oracle.cloudgen(np.full((h, w), 2.0, np.float32), bgr, grid, 2, 0.1, 5.0, 0)
    assert rec.shape[0] == (h // 2) * (w // 2)
    nrm = rec["normal"].reshape(h // 2, w // 2, 3)
    np.testing.assert_allclose(nrm[1:-1, 1:-1], np.broadcast_to([0, 0, -1], nrm[1:-1, 1:-1].shape), atol=1e-6)
    # the corner still has one valid neighbour pair, an isolated point has none -> zero normal
    assert np.isclose(np.linalg.norm(nrm[0, 0]), 1.0, atol=1e-6)
    depth = np.zeros((h, w), np.float32)
    depth[10, 10] = 1.0
    rec, _ = oracle.cloudgen(depth, bgr, grid, 2, 0.1, 5.0, 0)
    assert rec.shape[0] == 1 and (rec["normal"] == 0).all()
    # noisy scene: unit length or exactly zero
    fr = make_rgbd_frames(1, seed=3)[0]
    rec, _ = oracle.cloudgen(fr["depth"], fr["bgr"], _grid(oracle), 2, 0.1, 5.0, 1)
    ln = np.linalg.norm(rec["normal"].astype(np.float64), axis=1)
    assert (np.isclose(ln, 1.0, atol=1e-6) | (ln == 0)).all() and (ln == 0).sum() < rec.shape[0] // 20


# --------------------------------------------------------------------------- GPU
CASES = [
    dict(id="tum_step2", width=640, height=480, step=2, seed=0),
    dict(id="tum_step1", width=640, height=480, step=1, seed=1),
    dict(id="odd_step3", width=637, height=479, step=3, seed=2),
    dict(id="pitched", width=600, height=400, step=2, seed=3, pitched=True),
]


def _crop(fr, width, height):
    return np.ascontiguousarray(fr["depth"][:height, :width]), np.ascontiguousarray(fr["bgr"][:height, :width])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_hip_cloud_matches_oracle(oracle, case):
    import torch
    from plvs_amd import cloudgen
    w, h, step = case["width"], case["height"], case["step"]
    grid = cloudgen.InitCamGridPoints(w, h, step, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    np.testing.assert_array_equal(grid, _grid(oracle, step=step, width=w, height=h))
    gen = cloudgen.PointCloudGenerator(w, h, grid, step=step, min_depth=0.1, max_depth=5.0)
    for k, fr in enumerate(make_rgbd_frames(2, seed=case["seed"])):
        depth, bgr = _crop(fr, w, h)
        want, want_p2p = oracle.cloudgen(depth, bgr, grid, step, 0.1, 5.0, 100 + k)
        got, got_p2p = gen.GeneratePointCloudInCameraFrameBGRA(bgr, depth, 100 + k)
        assert got.shape[0] == want.shape[0] > 1000
        assert got.tobytes() == want.view(cloudgen.POINT_SURFEL).tobytes()
        np.testing.assert_array_equal(got_p2p, want_p2p)

        # device flavour, optionally with padded rows
        if case.get("pitched"):
            d_depth = torch.zeros((h, w + 24), dtype=torch.float32, device="cuda")[:, :w]
            d_bgr = torch.zeros((h, w + 8, 3), dtype=torch.uint8, device="cuda")[:, :w]
            d_depth.copy_(torch.from_numpy(depth))
            d_bgr.copy_(torch.from_numpy(bgr))
        else:
            d_depth, d_bgr = torch.from_numpy(depth).cuda(), torch.from_numpy(bgr).cuda()
        ng = gen.ngrid
        o = dict(d_xyz=torch.empty((ng, 3), dtype=torch.float32, device="cuda"),
                 d_rgb=torch.empty((ng, 3), dtype=torch.uint8, device="cuda"),
                 d_rgba=torch.empty((ng, 4), dtype=torch.uint8, device="cuda"),
                 d_kfid=torch.empty(ng, dtype=torch.int32, device="cuda"),
                 d_normals=torch.empty((ng, 3), dtype=torch.float32, device="cuda"),
                 d_point_depth=torch.empty(ng, dtype=torch.float32, device="cuda"),
                 d_pixel_to_point=torch.empty((h, w), dtype=torch.int32, device="cuda"))

        lib = cloudgen._lib
        import ctypes
        n = ctypes.c_int()
        lib.check(gen._lib.plvs_hip_cloudgen_generate_dev(
            gen._h, ctypes.c_void_p(d_depth.data_ptr()), d_depth.stride(0), ctypes.c_void_p(d_bgr.data_ptr()),
            d_bgr.stride(0), 0.1, 5.0, 100 + k, *[ctypes.c_void_p(o[key].data_ptr()) for key in
                                                 ("d_xyz", "d_rgb", "d_rgba", "d_kfid", "d_normals",
                                                  "d_point_depth", "d_pixel_to_point")],
            None, lib.current_stream_ptr(), ctypes.byref(n)))
        n = n.value
        assert n == want.shape[0]
        np.testing.assert_array_equal(o["d_xyz"][:n].cpu().numpy(), np.stack([want["x"], want["y"], want["z"]], -1))
        np.testing.assert_array_equal(o["d_rgb"][:n].cpu().numpy(), np.stack([want["r"], want["g"], want["b"]], -1))
        np.testing.assert_array_equal(o["d_rgba"][:n].cpu().numpy(),
                                      np.stack([want["r"], want["g"], want["b"], want["a"]], -1))
        assert (o["d_kfid"][:n].cpu().numpy() == 100 + k).all()
        assert o["d_normals"][:n].cpu().numpy().tobytes() == np.ascontiguousarray(want["normal"]).tobytes()
        np.testing.assert_array_equal(o["d_point_depth"][:n].cpu().numpy(), want["depth"])
        np.testing.assert_array_equal(o["d_pixel_to_point"].cpu().numpy(), want_p2p)


@pytest.mark.gpu
def test_hip_cloud_edge_cases(oracle):
    from plvs_amd import cloudgen
    w, h = 64, 48
    grid = cloudgen.InitCamGridPoints(w, h, 2, 50.0, 50.0, 31.5, 23.5)
    gen = cloudgen.PointCloudGenerator(w, h, grid, step=2, min_depth=0.1, max_depth=5.0)
    bgr = np.arange(h * w * 3, dtype=np.uint32).astype(np.uint8).reshape(h, w, 3)
    for depth in (np.zeros((h, w), np.float32),                      # nothing valid
                  np.full((h, w), np.nan, np.float32),
                  np.full((h, w), 5.0, np.float32),                   # d < max is strict
                  np.full((h, w), np.float32(0.1), np.float32),       # float 0.1 > double 0.1 -> all valid
                  np.full((h, w), 2.0, np.float32)):
        want, want_p2p = oracle.cloudgen(depth, bgr, grid, 2, 0.1, 5.0, 3)
        got, got_p2p = gen.GeneratePointCloudInCameraFrameBGRA(bgr, depth, 3)
        assert got.shape[0] == want.shape[0]
        assert got.tobytes() == want.view(cloudgen.POINT_SURFEL).tobytes()
        np.testing.assert_array_equal(got_p2p, want_p2p)
    assert oracle.cloudgen(np.full((h, w), np.float32(0.1), np.float32), bgr, grid, 2, 0.1, 5.0, 3)[0].shape[0] == 768
    with pytest.raises(ValueError):
        gen.GeneratePointCloudInCameraFrameBGRA(bgr[:10], np.zeros((10, w), np.float32), 0)


@pytest.mark.gpu
def test_depth_to_map_chain_matches_oracle(oracle):
    """depth image -> cloud -> chisel integrate, all resident in HBM, against the two oracles chained."""
    import torch
    from plvs_amd import cloudgen
    from plvs_amd.tsdf import TsdfChisel
    w, h = TUM1["width"], TUM1["height"]
    grid = cloudgen.InitCamGridPoints(w, h, 2, TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"])
    gen = cloudgen.PointCloudGenerator(w, h, grid, step=2, min_depth=0.1, max_depth=5.0)
    frames = make_rgbd_frames(3, seed=5)
    ref = oracle.chisel(0.05)
    hip = TsdfChisel(0.05)
    ng = gen.ngrid
    d_xyz = torch.empty((3 * ng, 3), dtype=torch.float32, device="cuda")
    d_rgb = torch.empty((3 * ng, 3), dtype=torch.uint8, device="cuda")
    d_kfid = torch.empty(3 * ng, dtype=torch.int32, device="cuda")
    offsets = [0]
    for k, fr in enumerate(frames):
        rec, _ = oracle.cloudgen(fr["depth"], fr["bgr"], grid, 2, 0.1, 5.0, k)
        ref.integrate(np.stack([rec["x"], rec["y"], rec["z"]], -1), np.stack([rec["r"], rec["g"], rec["b"]], -1),
                      rec["kfid"], fr["Twc"])
        o = offsets[-1]
        n = gen.generate_dev(torch.from_numpy(fr["bgr"]).cuda(), torch.from_numpy(fr["depth"]).cuda(), k,
                             d_xyz[o:], d_rgb=d_rgb[o:], d_kfid=d_kfid[o:])
        assert n == rec.shape[0]
        offsets.append(o + n)
    d_Twc = torch.from_numpy(np.stack([fr["Twc"] for fr in frames])).cuda()
    hip.integrate_batch_dev(d_xyz, d_rgb, d_kfid, np.array(offsets, np.int32), d_Twc)
    torch.cuda.synchronize()
    ids = hip.chunk_ids()
    assert sorted(map(tuple, ids)) == sorted(map(tuple, ref.chunk_ids()))
    for cid in ids[:: max(1, len(ids) // 40)]:
        a, b = hip.get_chunk(*cid), ref.get_chunk(*cid)
        for x, y in zip(a, b):
            assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
