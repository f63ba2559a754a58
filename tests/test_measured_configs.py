"""Parity at the configurations bench.py measures and BASELINE.json names (the other GPU tests use
reduced image sizes so that the oracle finishes in seconds):

  configs[2]  the exact bench step — 100 full-resolution keyframes in ONE integrate_batch_dev on the
              6x4x3 m room, chisel 5 cm / 5 m — in the bit-exact and in the order-free mode, first on an
              empty map and again on the populated one; carving after integrate at 640x480.
              The STREAMING headline of round 4 — the first two 100-key-frame steps of the office loop, as bench.py runs
              them — in both modes (order-free: the stated tolerance against the exact mean AND against the reference).
  configs[3]  voxblox 2 cm in the 16x12x3 m room at full resolution, points beyond the wrapper's 5 m
              ray limit, carving off and on.
  configs[4]  KITTI-sized stereo pair -> SGM disparity -> depth (bf 386.1448) -> cloud -> chisel 10 cm,
              against the chained oracles.
"""
import numpy as np
import pytest

from tests import oracle_lib  # noqa: F401  (the `oracle` fixture lives in conftest)
from tests.plvs_amd_synth import make_keyframes, TUM1
from tests.test_tsdf_chisel import (ORDER_FREE_SDF_ATOL, ORDER_FREE_WEIGHT_RTOL, _depth_image, compare_maps)

pytestmark = pytest.mark.gpu
HEADLINE_STEPS = 5


def _snapshot(m):
    return {tuple(c): tuple(np.array(x, copy=True) for x in m.get_chunk(*c)) for c in m.chunk_ids()}


@pytest.fixture(scope="module")
def bench_stream(oracle):
    """The bench's input (bench.py: make_keyframes(100, max_depth=5.0, seed=0)) and the oracle's map after
    one and after two passes over it."""
    kfs = make_keyframes(100, max_depth=5.0, seed=0)
    ora = oracle.chisel(0.05)
    visits = []
    snaps = []
    for _ in range(2):
        v = 0
        for kf in kfs:
            ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
            v += ora.last_visits()
        visits.append(v)
        snaps.append(_snapshot(ora))
    return kfs, visits, snaps


@pytest.mark.parametrize("order_free", [False, True])
def test_chisel_bench_step_matches_oracle(bench_stream, order_free):
    import torch
    from plvs_amd.tsdf import TsdfChisel
    kfs, visits, snaps = bench_stream
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    dev = TsdfChisel(0.05, max_chunks=16384, order_free=order_free)      # as bench.py creates it
    for step in range(2):
        dev.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
        torch.cuda.synchronize()
        st = dev.last_stats()
        assert st["visits"] == visits[step] and st["points"] == xyz.shape[0]
        want = snaps[step]
        assert {tuple(c) for c in dev.chunk_ids()} == set(want)
        worst_s = worst_w = 0.0
        for cid, a in want.items():
            b = dev.get_chunk(*cid)
            assert np.array_equal(a[2], b[2]), f"kfid differs in chunk {cid}"
            assert np.array_equal(a[3], b[3]), f"colour differs in chunk {cid}"
            if order_free:
                known = a[1] > 0
                assert np.array_equal(known, b[1] > 0)
                if known.any():
                    worst_s = max(worst_s, float(np.abs(a[0][known] - b[0][known]).max()))
                    worst_w = max(worst_w, float((np.abs(a[1][known] - b[1][known]) / a[1][known]).max()))
            else:
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"sdf differs in chunk {cid}"
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), f"weight differs in chunk {cid}"
        if order_free:
            print(f"step {step}: order-free deviations sdf {worst_s:.3g} m, weight {worst_w:.3g} rel")
            assert worst_s <= ORDER_FREE_SDF_ATOL and worst_w <= ORDER_FREE_WEIGHT_RTOL
    dev.close()


@pytest.fixture(scope="module")
def headline_stream(oracle):
    """The first HEADLINE_STEPS steps of bench.py's STREAMING headline (key frames 0-99 into an empty map, 100-199 into that
    map, ...) on the oracle: voxel planes after each, and the exact (f64) mean of every voxel's visits beside the reference's
    f32 running mean.  Five steps: the first call on an empty map, the calls in which the walk's first-pass table follows the
    scene (2048 -> 1024 entries for the 2-D tiles of the depth entry point), revisits of the desk island."""
    from tests.synth_scene import make_stream_keyframes
    kfs = make_stream_keyframes(100 * HEADLINE_STEPS, max_depth=5.0, seed=0, threads=16, images=True)
    ora = oracle.chisel(0.05)
    ora.track_exact()
    visits, snaps, exact = [], [], []
    for step in range(HEADLINE_STEPS):
        v = 0
        for kf in kfs[100 * step:100 * step + 100]:
            ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
            v += ora.last_visits()
        visits.append(v)
        snaps.append(_snapshot(ora))
        exact.append({tuple(c): tuple(np.array(x, copy=True) for x in ora.get_chunk_exact(*c)) for c in ora.chunk_ids()})
    return kfs, visits, snaps, exact


def _depth_batch(g, max_depth=5.0, step=2):
    """The key frames as bench.py hands them to the depth entry point: 640 x 480 images, depth and colour at the pixels of
    the stride-2 grid."""
    import torch
    gh, gw = g[0]["depth_grid"].shape
    d = torch.zeros((len(g), gh * step, gw * step), dtype=torch.float32, device="cuda")
    c = torch.zeros((len(g), gh * step, gw * step, 3), dtype=torch.uint8, device="cuda")
    d[:, ::step, ::step] = torch.from_numpy(np.stack([k["depth_grid"] for k in g])).cuda()
    c[:, ::step, ::step] = torch.from_numpy(np.stack([k["rgb_grid"] for k in g])).cuda()
    return (d, c, torch.from_numpy(g[0]["cam_grid"]).cuda(), step, 0.1, max_depth,
            torch.from_numpy(np.array([int(k["kfid"][0]) for k in g], np.int32)).cuda(),
            torch.from_numpy(np.stack([k["Twc"] for k in g])).cuda())


@pytest.mark.parametrize("order_free,entry", [(False, "cloud"), (True, "cloud"), (True, "depth")])
def test_chisel_streaming_headline_steps_match_oracle(headline_stream, order_free, entry):
    """The headline as the bench runs it since round 4: tiles of walls 3-5 m away (800-2 000 voxels each: the 2048-entry
    first pass, the 4096-entry pass over what overflowed, the general kernel behind it), chunk allocation and colour folds
    inside the step.  Ordered mode: bit for bit.  Order-free mode: kfid and colour exact; sdf / weight within 2e-5 m / 5e-5
    of the EXACT mean of the reference's own visits, and within the stated bound of the reference's f32 map — a voxel with n
    visits: 2e-5 m + n 2^-24 tau, 5e-5 + n 2^-24 (tau = the truncation distance at 5 m; bench.py asserts the same)."""
    import torch
    from plvs_amd.tsdf import TsdfChisel
    kfs, visits, snaps, exact = headline_stream
    tau = max((0.0019 * 25.0 - 0.00152 * 5.0 + 0.001504) * 6.0, 2.0 * np.sqrt(3.0) * 0.05)
    w_min = 1.0 / (2.0 * tau)
    dev = TsdfChisel(0.05, max_chunks=16384, order_free=order_free)
    for step in range(HEADLINE_STEPS if order_free else 2):      # (the ordered mode: two steps, as in round 4)
        g = kfs[100 * step:100 * step + 100]
        if entry == "depth":      # the headline of round 5: GeneratePointCloudInCameraFrameBGRA + InsertCloud in one call
            dev.integrate_depth_batch_dev(*_depth_batch(g))
            torch.cuda.synchronize()
            assert dev.last_stats()["visits"] == visits[step]
        else:
            xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in g])).cuda()
            rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in g])).cuda()
            kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in g]).astype(np.int32)).cuda()
            Twc = torch.from_numpy(np.stack([k["Twc"] for k in g])).cuda()
            offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in g]).astype(np.int32)
            dev.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
            torch.cuda.synchronize()
            st = dev.last_stats()
            assert st["visits"] == visits[step] and st["points"] == xyz.shape[0]
        want = snaps[step]
        assert {tuple(c) for c in dev.chunk_ids()} == set(want)
        assert len(want) > 80
        ws = ww = bound_s = bound_w = 0.0
        for cid, a in want.items():
            b = dev.get_chunk(*cid)
            assert np.array_equal(a[2], b[2]), f"kfid differs in chunk {cid}"
            assert np.array_equal(a[3], b[3]), f"colour differs in chunk {cid}"
            if order_free:
                known = a[1] > 0
                assert np.array_equal(known, b[1] > 0)
                if known.any():
                    xs, xw = exact[step][cid]
                    ws = max(ws, float(np.abs(b[0][known] - xs[known]).max()))
                    ww = max(ww, float((np.abs(b[1][known] - xw[known]) / xw[known]).max()))
                    n_up = np.ceil(xw[known] / w_min) + 1.0
                    bound_s = max(bound_s, float((np.abs(a[0][known] - b[0][known]) / (2e-5 + n_up * 2.0 ** -24 * tau)).max()))
                    bound_w = max(bound_w, float((np.abs(a[1][known] - b[1][known]) / a[1][known] / (5e-5 + n_up * 2.0 ** -24)).max()))
            else:
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"sdf differs in chunk {cid}"
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), f"weight differs in chunk {cid}"
        if order_free:
            assert ws <= ORDER_FREE_SDF_ATOL and ww <= ORDER_FREE_WEIGHT_RTOL, (ws, ww)
            assert bound_s <= 1.0 and bound_w <= 1.0, (bound_s, bound_w)
    dev.close()


def test_chisel_streaming_headline_later_steps_depth_entry_equals_point_streams():
    """Steps 1-13 of the headline (1 300 key frames: the map grows to ~450 chunks, the walk's first-pass table has settled,
    the colour chain runs on predicted sizes where the run count allows): the map of the depth entry point against the map
    of the point-stream entry point, both order-free, bit for bit after steps 5, 9 and 13.  (The point-stream map of ALL 25
    steps is compared with the oracle inside bench.py, `parity`; the first five steps against the oracle above.)"""
    import torch
    from tests.synth_scene import make_stream_keyframes
    from plvs_amd.tsdf import TsdfChisel
    a = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    b = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    for step in range(13):
        g = make_stream_keyframes(100, first=100 * step, max_depth=5.0, seed=0, threads=16, images=True)
        a.integrate_depth_batch_dev(*_depth_batch(g))
        b.integrate_batch_dev(torch.from_numpy(np.concatenate([k["xyz"] for k in g])).cuda(),
                              torch.from_numpy(np.concatenate([k["rgb"] for k in g])).cuda(),
                              torch.from_numpy(np.concatenate([k["kfid"] for k in g]).astype(np.int32)).cuda(),
                              np.cumsum([0] + [k["xyz"].shape[0] for k in g]).astype(np.int32),
                              torch.from_numpy(np.stack([k["Twc"] for k in g])).cuda())
        assert a.last_stats()["visits"] == b.last_stats()["visits"] > 10_000_000
        if step in (4, 8, 12):
            assert compare_maps(a, b) > 150
    a.close()
    b.close()


@pytest.mark.parametrize("kind", ["far", "mixed"])
def test_chisel_full_resolution_carving_after_integrate(oracle, kind):
    """configs[2] 'carving off then on': five 640x480 keyframes, then CarveWithDepth with a full-size depth
    image from two of the poses, then one more keyframe on the carved map."""
    from plvs_amd.tsdf import TsdfChisel
    cam = dict(TUM1)
    kfs = make_keyframes(6, max_depth=5.0, seed=41)
    ora = oracle.chisel(0.05)
    dev = TsdfChisel(0.05, max_chunks=16384)
    for kf in kfs[:5]:
        ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        dev.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    for step, pose in enumerate((kfs[1]["Twc"], kfs[4]["Twc"])):
        depth = _depth_image(cam, kind, seed=step)
        assert depth.shape == (480, 640)
        want_n, want_ids = ora.carve(depth, cam["fx"], cam["fy"], cam["cx"], cam["cy"], pose)
        assert dev.carve(depth, cam["fx"], cam["fy"], cam["cx"], cam["cy"], pose) == want_n
        assert {tuple(i) for i in dev.updated_chunk_ids()} == {tuple(i) for i in want_ids}
        compare_maps(ora, dev)
    ora.integrate(kfs[5]["xyz"], kfs[5]["rgb"], kfs[5]["kfid"], kfs[5]["Twc"])
    dev.integrate(kfs[5]["xyz"], kfs[5]["rgb"], kfs[5]["kfid"], kfs[5]["Twc"])
    assert compare_maps(ora, dev) > 20
    dev.close()


@pytest.mark.parametrize("carving,nkf", [(False, 6), (True, 2)])
def test_voxblox_2cm_large_room_full_resolution(oracle, carving, nkf):
    """configs[3]: the bench's voxblox input (16x12x3 m room, depths to 8 m: many points lie beyond the
    wrapper's 5 m ray limit and are dropped without carving / become clearing rays with it), 2 cm voxels,
    full-resolution keyframes, one batch."""
    import torch
    from plvs_amd.tsdf import TsdfVoxblox
    from tests.test_tsdf_voxblox import compare, rgba_of
    kfs = make_keyframes(nkf, room_size=(16.0, 12.0, 3.0), max_depth=8.0, seed=0)
    assert max(float(k["xyz"][:, 2].max()) for k in kfs) > 5.0
    ora = oracle.voxblox(0.02, carving=carving)
    v = 0
    for k in kfs:
        ora.integrate(k["xyz"], rgba_of(k), k["Twc"])
        v += ora.last_visits()
    dev = TsdfVoxblox(0.02, use_carving=carving, max_blocks=65536)
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgba = torch.from_numpy(np.concatenate([rgba_of(k) for k in kfs])).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    dev.integrate_batch_dev(xyz, rgba, offsets, Twc)
    torch.cuda.synchronize()
    assert dev.last_stats()["visits"] == v
    assert compare(ora, dev) > 50
    dev.close()


def test_kitti_chain_sgm_depth_cloud_chisel_10cm(oracle):
    """configs[4]: 1240x376 pair -> semi-global matching -> depth = bf / d (bf = 386.1448,
    Examples_old/Stereo/KITTI00-02.yaml:45) -> cloud with normals -> chisel TSDF 10 cm; every link is the
    device path, compared with the oracles chained the same way."""
    import torch
    from plvs_amd import cloudgen
    from tests.pgm import golden_frame
    from plvs_amd.sgm import StereoSGM
    from plvs_amd.tsdf import TsdfChisel
    left = np.ascontiguousarray(golden_frame("urban1_1241x376.pgm")[:, :1240])
    right = np.ascontiguousarray(golden_frame("urban1_right_1241x376.pgm")[:, :1240])
    h, w = left.shape
    fx, fy, cx, cy, bf = 718.856, 718.856, 607.1928, 185.2157, 386.1448      # KITTI00-02.yaml
    # --- disparity
    sgm = StereoSGM(w, h)
    d_disp = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    sgm.execute_dev(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), d_disp)
    torch.cuda.synchronize()
    disp_dev = d_disp.cpu().numpy()
    disp_ora = oracle.sgm(left, right)
    assert np.array_equal(disp_dev, disp_ora)

    def depth_of(disp):
        d = np.zeros(disp.shape, np.float32)
        ok = disp > 0
        d[ok] = np.float32(bf) / disp[ok].astype(np.float32)
        return d

    bgr = np.repeat(left[:, :, None], 3, axis=2)
    grid = cloudgen.InitCamGridPoints(w, h, 2, fx, fy, cx, cy)
    gen = cloudgen.PointCloudGenerator(w, h, grid, step=2, min_depth=0.5, max_depth=20.0)
    Twc = np.eye(4, dtype=np.float32)[:3]
    # --- oracle chain
    rec, _ = oracle.cloudgen(depth_of(disp_ora), bgr, grid, 2, 0.5, 20.0, 7)
    ref = oracle.chisel(0.10)
    ref.integrate(np.stack([rec["x"], rec["y"], rec["z"]], -1), np.stack([rec["r"], rec["g"], rec["b"]], -1),
                  rec["kfid"], Twc)
    # --- device chain, resident in HBM
    ng = gen.ngrid
    d_xyz = torch.empty((ng, 3), dtype=torch.float32, device="cuda")
    d_rgb = torch.empty((ng, 3), dtype=torch.uint8, device="cuda")
    d_kfid = torch.empty(ng, dtype=torch.int32, device="cuda")
    n = gen.generate_dev(torch.from_numpy(bgr).cuda(), torch.from_numpy(depth_of(disp_dev)).cuda(), 7, d_xyz,
                         d_rgb=d_rgb, d_kfid=d_kfid)
    assert n == rec.shape[0] > 20000
    hip = TsdfChisel(0.10, max_chunks=16384)
    hip.integrate_batch_dev(d_xyz, d_rgb, d_kfid, np.array([0, n], np.int32), torch.from_numpy(Twc[None]).cuda())
    torch.cuda.synchronize()
    assert hip.last_stats()["visits"] == ref.last_visits()
    assert compare_maps(ref, hip) > 10
    hip.close()


@pytest.mark.gpu
def test_kitti_chain_as_shipped_libelas_depth_cloud_chisel_10cm(oracle):
    """configs[4] as the shipped YAML runs it (Examples_old/Stereo/KITTI00-02.yaml: libelas, skDownsampleStep 2 ->
    subsampling): 1241x376 pair -> libelas::Elas::process with EVERY device stage — descriptors, support candidates, the
    two computeDisparity calls leaving their maps in HBM, then leftRightConsistencyCheck / removeSmallSegments /
    gapInterpolation / adaptiveMean as ONE call on those maps (plvs_hip_elas_postprocess) — -> ProcessStereoLibelas'
    disparity -> depth on the device (src/PointCloudKeyFrame.cc:399-420, bf = 386.1448) -> cloud with normals -> chisel
    TSDF 10 cm.  Against: the compiled reference pipeline's maps (bit for bit), then the same conversion in numpy and the
    chained oracles (cloud records byte for byte, map bit for bit)."""
    import torch
    from tests import elas_ref
    from tests.test_elas import pair
    if not elas_ref.available():
        pytest.skip("needs oracle/_ref/libelas_ref.so (built where /root/reference is)")
    from plvs_amd import cloudgen
    from plvs_amd.elas import ElasGPU
    from plvs_amd.tsdf import TsdfChisel
    left, right = pair()
    h, w = left.shape
    fx, fy, cx, cy, bf = 718.856, 718.856, 607.1928, 185.2157, 386.1448      # KITTI00-02.yaml
    want1, want2 = elas_ref.reference(left, right, subsampling=True, plvs=True)
    e = ElasGPU(ElasGPU.Parameters(subsampling=True))
    e.setImages(left, right)
    done = {}

    def disparity(a):      # (the pipeline wants the map back: the copy in HBM is what the chain below reads)
        return e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, a["right_image"], w, h)

    def lr(D1, D2):        # the first post-processing hook: the whole chain in one call, the later hooks have nothing left to do
        done["maps"] = e.postProcess(w, h, postprocess_only_left=True, filter_adaptive_mean=True)
        D1[:], D2[:] = done["maps"]

    got1, got2 = elas_ref.run_with(left, right, disparity, lambda D, ww, hh, sub: D, subsampling=True, plvs=True,
                                   support_candidates=lambda a: e.supportCandidates(None, None, w, h),
                                   post=dict(left_right_check=lr, remove_small_segments=lambda D: None,
                                             gap_interpolation=lambda D: None))
    assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)) and np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
    assert (want1 >= 0).mean() > 0.3
    # --- depth as ProcessStereoLibelas makes it
    depth_ref = np.zeros((h, w), np.float32)
    with np.errstate(divide="ignore"):
        q = (np.float32(bf) / want1).astype(np.float32)
    depth_ref[0:2 * want1.shape[0]:2, 0:2 * want1.shape[1]:2] = q
    depth_ref[0:2 * want1.shape[0]:2, 1:2 * want1.shape[1]:2] = q
    d_depth = torch.empty((h, w), dtype=torch.float32, device="cuda")
    e.depthDev(bf, 2, d_depth)
    torch.cuda.synchronize()
    assert np.array_equal(d_depth.cpu().numpy().view(np.uint32), depth_ref.view(np.uint32))
    # --- cloud + TSDF
    bgr = np.repeat(left[:, :, None], 3, axis=2)
    grid = cloudgen.InitCamGridPoints(w, h, 2, fx, fy, cx, cy)
    gen = cloudgen.PointCloudGenerator(w, h, grid, step=2, min_depth=0.5, max_depth=20.0)
    Twc = np.eye(4, dtype=np.float32)[:3]
    rec, _ = oracle.cloudgen(depth_ref, bgr, grid, 2, 0.5, 20.0, 7)
    ref = oracle.chisel(0.10)
    ref.integrate(np.stack([rec["x"], rec["y"], rec["z"]], -1), np.stack([rec["r"], rec["g"], rec["b"]], -1), rec["kfid"], Twc)
    ng = gen.ngrid
    d_xyz = torch.empty((ng, 3), dtype=torch.float32, device="cuda")
    d_rgb = torch.empty((ng, 3), dtype=torch.uint8, device="cuda")
    d_kfid = torch.empty(ng, dtype=torch.int32, device="cuda")
    n = gen.generate_dev(torch.from_numpy(bgr).cuda(), d_depth, 7, d_xyz, d_rgb=d_rgb, d_kfid=d_kfid)
    assert n == rec.shape[0] > 10000
    hip = TsdfChisel(0.10, max_chunks=16384)
    hip.integrate_batch_dev(d_xyz, d_rgb, d_kfid, np.array([0, n], np.int32), torch.from_numpy(Twc[None]).cuda())
    torch.cuda.synchronize()
    assert hip.last_stats()["visits"] == ref.last_visits()
    assert compare_maps(ref, hip) > 10
    hip.close()
    e.close()
