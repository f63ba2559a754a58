"""libelas: the two methods the reference's own GPU build overrides (class ElasGPU : public Elas,
Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45) — Elas::computeDisparity (CPU/elas.cpp:840-968) and Elas::adaptiveMean
(:1349-1572) — as PointCloudKeyFrame::ProcessStereoLibelas reaches them (src/PointCloudKeyFrame.cc:335-432).

The reference's CPU sources run here compiled unmodified (oracle/_ref/libelas_ref.so) with hooks in ElasGPU's two places
(tests/elas_ref.py): the arguments are the ones the reference pipeline itself produces for a real stereo pair
(tests/golden/urban1*: the tree's own input, cropped).  CPU tests pin oracle/elas.c against the compiled methods; GPU tests
put the HIP path through the C ABI against the oracle, and INTO the reference pipeline against the pure reference run.
Where oracle/_ref is absent the committed capture tests/golden/elas_capture.npz stands in (scripts/make_elas_golden.py)."""
import os

import numpy as np
import pytest

from tests import elas_ref
from tests.pgm import read_pgm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
needs_ref = pytest.mark.skipif(not elas_ref.available(), reason="needs oracle/_ref/libelas_ref.so (built where /root/reference is)")


def pair(width=None, height=None):
    left = read_pgm(os.path.join(GOLDEN, "urban1_1241x376.pgm"))
    right = read_pgm(os.path.join(GOLDEN, "urban1_right_1241x376.pgm"))
    h, w = left.shape
    width, height = width or w, height or h
    return np.ascontiguousarray(left[:height, :width]), np.ascontiguousarray(right[:height, :width])


def golden_capture():
    z = np.load(os.path.join(GOLDEN, "elas_capture.npz"))
    calls = []
    for i in range(int(z["n_disparity_calls"])):
        calls.append({k: z[f"d{i}_{k}"] for k in ("support", "tri", "grid", "grid_dims", "D")})
        calls[-1]["I1_desc"], calls[-1]["I2_desc"] = z["I1_desc"], z["I2_desc"]
        calls[-1]["support"] = np.ascontiguousarray(calls[-1]["support"]).view(elas_ref.SUPPORT).reshape(-1)
        calls[-1]["tri"] = np.ascontiguousarray(calls[-1]["tri"]).view(elas_ref.TRIANGLE).reshape(-1)
        for k in ("right_image", "width", "height", "subsampling"):
            calls[-1][k] = int(z[f"d{i}_{k}"])
    means = [dict(D_in=z[f"m{i}_D_in"], D_out=z[f"m{i}_D_out"], width=int(z[f"m{i}_width"]), height=int(z[f"m{i}_height"]),
                  subsampling=int(z[f"m{i}_subsampling"])) for i in range(int(z["n_mean_calls"]))]
    return calls, means


# ------------------------------------------------------------------ the oracle against the reference's compiled methods
@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300), (333, 201)])
def test_oracle_compute_disparity_and_adaptive_mean_equal_the_reference_source(oracle, subsampling, size):
    left, right = pair(*size)
    disp_calls, mean_calls, _ = elas_ref.capture(left, right, subsampling=subsampling, plvs=False)
    assert len(disp_calls) == 2 and len(mean_calls) == 2      # left and right image; both post-processed
    for a in disp_calls:
        got = oracle.elas_compute_disparity(a)
        assert np.array_equal(got.reshape(-1).view(np.uint32), a["D"].view(np.uint32)), f"right_image={a['right_image']}"
        assert (a["D"] >= 0).mean() > 0.3                      # a real disparity map, not an empty one
    for m in mean_calls:
        got = oracle.elas_adaptive_mean(m["D_in"], m["width"], m["height"], m["subsampling"])
        assert np.array_equal(got.reshape(-1).view(np.uint32), m["D_out"].view(np.uint32))
        assert not np.array_equal(m["D_in"], m["D_out"])


@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
def test_reference_pipeline_with_the_oracle_in_elasgpus_place_equals_the_reference(oracle, subsampling):
    left, right = pair(800, 376)
    want = elas_ref.reference(left, right, subsampling=subsampling, plvs=True)     # PLVS: postprocess_only_left
    got = elas_ref.run_with(left, right, oracle.elas_compute_disparity, oracle.elas_adaptive_mean, subsampling=subsampling,
                            plvs=True)
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    assert (want[0] >= 0).mean() > 0.3


@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300)])
def test_reference_pipeline_with_the_oracles_candidate_grid_gives_the_references_support_points(oracle, subsampling, size):
    """The candidate loop of Elas::computeSupportMatches (elas.cpp:434-456) replaced by the oracle's, the reference's own
    filters after it: the support points and triangles the pipeline then hands to computeDisparity are the reference's."""
    left, right = pair(*size)
    want_calls, _, want = elas_ref.capture(left, right, subsampling=subsampling, plvs=True)
    got_calls = []

    def record(a):
        got_calls.append(a)
        return oracle.elas_compute_disparity(a)
    got = elas_ref.run_with(left, right, record, oracle.elas_adaptive_mean, subsampling=subsampling, plvs=True,
                            support_candidates=oracle.elas_support_candidates)
    assert len(got_calls) == 2
    for g, w in zip(got_calls, want_calls):
        assert len(w["support"]) > 100
        assert np.array_equal(g["support"], w["support"]) and np.array_equal(g["tri"], w["tri"])
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("middlebury", [False, True])
@pytest.mark.parametrize("subsampling", [False, True])
def test_reference_pipeline_with_the_oracles_post_processing_equals_the_reference(oracle, subsampling, middlebury):
    """leftRightConsistencyCheck, removeSmallSegments and gapInterpolation replaced by the oracle's, with the ROBOTICS
    parameters PLVS starts from and with the MIDDLEBURY set (add_corners, gaps of 5000 pixels, speckle similarity 1): the
    pipeline's maps — and every intermediate map, taken right after each stage — are the reference's."""
    left, right = pair(900, 376)
    mode = (2 if middlebury else 0)          # both images post-processed: exercises every method on two maps
    prm = dict(lr=2, size=200, sim=1.0, gap=5000 if middlebury else 3, corners=middlebury)     # elas.h:97-155
    stages_ref, stages_got = [], []
    want = elas_ref._run(left, right, subsampling, mode, None, None, post=None)

    def post(record, use_oracle):
        def lr(D1, D2):
            if use_oracle:
                oracle.elas_left_right_check(D1, D2, subsampling, prm["lr"])
            record.append(("lr", D1.copy(), D2.copy()))

        def seg(D):
            if use_oracle:
                oracle.elas_remove_small_segments(D, subsampling, prm["size"], prm["sim"])
            record.append(("seg", D.copy()))

        def gap(D):
            if use_oracle:
                oracle.elas_gap_interpolation(D, subsampling, prm["gap"], prm["corners"])
            record.append(("gap", D.copy()))
        return dict(left_right_check=lr, remove_small_segments=seg, gap_interpolation=gap)
    got = elas_ref._run(left, right, subsampling, mode, None, None, post=post(stages_got, True))
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    assert [s[0] for s in stages_got] == ["lr", "seg", "seg", "gap", "gap"]
    assert (want[0] >= 0).mean() > 0.3


@needs_ref
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300), (333, 201), (64, 48)])
def test_oracle_descriptor_equals_the_reference_source(oracle, half, size):
    """libelas::Descriptor (Sobel over the flat buffer + the 16 samples) on both images of the pair: widths that are and are
    not multiples of 16, full and half resolution; and the descriptors the pipeline itself hands to computeDisparity."""
    import ctypes
    lib = ctypes.CDLL(elas_ref.REF)
    lib.ref_elas_descriptor.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    for img in pair(*size):
        h, w = img.shape
        want = np.zeros(16 * w * h, np.uint8)
        lib.ref_elas_descriptor(img.ctypes.data, w, h, w, int(half), want.ctypes.data)
        got = oracle.elas_descriptor(img, half)
        assert np.array_equal(got, want), f"{int((got != want).sum())} bytes differ"
        assert len(np.unique(got)) > 50
    if size == (640, 300):
        left, right = pair(*size)
        calls, _, _ = elas_ref.capture(left, right, subsampling=half, plvs=True)
        assert np.array_equal(oracle.elas_descriptor(left, half), calls[0]["I1_desc"])
        assert np.array_equal(oracle.elas_descriptor(right, half), calls[0]["I2_desc"])


TREE_INPUT = "/root/reference/Thirdparty/libelas-gpu/input"


@needs_ref
@pytest.mark.skipif(not os.path.isdir(TREE_INPUT), reason="the reference tree's other stereo pairs are not committed (1 MB images)")
@pytest.mark.parametrize("name,subsampling", [("cones", False), ("cones", True), ("aloe", True), ("urban3", False)])
def test_every_oracle_stage_on_the_trees_other_pairs(oracle, name, subsampling):
    """Dev-container test: the reference tree's own input pairs beyond urban1 (cones 900 x 750, aloe 1282 x 1110,
    urban3 1344 x 391 — other textures, disparity ranges up to the 255 limit, widths that are not multiples of 16).  The
    descriptors against libelas::Descriptor; then the pipeline with the oracle's candidate grid, computeDisparity,
    left/right check, speckle removal, gap interpolation and adaptive mean in place of the reference's methods: the very
    maps of the pure reference run."""
    import ctypes

    def pgm(path):       # (binary P5 with comment lines, which GIMP writes into some of the tree's files)
        data = open(path, "rb").read()
        tokens, pos = [], 0
        while len(tokens) < 4:
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            tokens.append(data[pos:end])
            pos = end
        assert tokens[0] == b"P5" and int(tokens[3]) == 255
        w_, h_ = int(tokens[1]), int(tokens[2])
        return np.frombuffer(data, np.uint8, w_ * h_, pos + 1).reshape(h_, w_).copy()
    left, right = pgm(f"{TREE_INPUT}/{name}_left.pgm"), pgm(f"{TREE_INPUT}/{name}_right.pgm")
    lib = ctypes.CDLL(elas_ref.REF)
    lib.ref_elas_descriptor.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    h, w = left.shape
    want_desc = np.zeros(16 * w * h, np.uint8)
    lib.ref_elas_descriptor(left.ctypes.data, w, h, w, int(subsampling), want_desc.ctypes.data)
    assert np.array_equal(oracle.elas_descriptor(left, subsampling), want_desc)
    want = elas_ref.reference(left, right, subsampling=subsampling, plvs=True)
    post = dict(left_right_check=lambda D1, D2: oracle.elas_left_right_check(D1, D2, subsampling),
                remove_small_segments=lambda D: oracle.elas_remove_small_segments(D, subsampling),
                gap_interpolation=lambda D: oracle.elas_gap_interpolation(D, subsampling))
    got = elas_ref.run_with(left, right, oracle.elas_compute_disparity, oracle.elas_adaptive_mean, subsampling=subsampling,
                            plvs=True, support_candidates=oracle.elas_support_candidates, post=post)
    for g, wv in zip(got, want):
        assert np.array_equal(g.view(np.uint32), wv.view(np.uint32))
    assert (want[0] >= 0).mean() > 0.3


@needs_ref
def test_adaptive_mean_alone_on_synthetic_maps_equals_the_reference_source(oracle):
    """Ramps, steps of 2 / 4 / 8 / 16 levels (the exponent classes of the subsampling branch's mask), invalid islands,
    borders: Elas::adaptiveMean alone.  (Maps of at least 32 KB: the reference reads its scratch image where it never
    wrote it — column 3 at full resolution, rows 0-2 — and only a block that large is fresh zero pages under the
    wrapper's mmap threshold; smaller ones come back from the heap with whatever the process left there.)"""
    import ctypes
    lib = ctypes.CDLL(elas_ref.REF)
    lib.ref_elas_adaptive_mean.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(5)
    for sub in (0, 1):
        for (w, h) in ((128, 80), (131, 77), (200, 45)):
            w, h = (2 * w + 1, 2 * h) if sub else (w, h)
            W, H = (w // 2, h // 2) if sub else (w, h)
            D = (np.linspace(0, 60, W)[None, :] + np.linspace(0, 9, H)[:, None]).astype(np.float32)
            D += rng.choice([0, 0, 0, 2, 4, 8, 16, 33.5], size=D.shape).astype(np.float32)
            D[rng.random(D.shape) < 0.15] = -1.0
            D[H // 3:H // 3 + 3, W // 4:W // 2] = -10.0
            want = D.copy()
            lib.ref_elas_adaptive_mean(want.ctypes.data, w, h, sub)
            got = oracle.elas_adaptive_mean(D, w, h, sub)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (sub, w, h)


def test_oracle_reproduces_the_committed_capture_of_the_reference(oracle):
    """Runs everywhere: the arguments and results of the reference's methods on a 256 x 128 crop, captured by
    scripts/make_elas_golden.py from the compiled reference."""
    calls, means = golden_capture()
    assert len(calls) == 2 and len(means) == 2 and {m["subsampling"] for m in means} == {0, 1}
    for a in calls:
        got = oracle.elas_compute_disparity(a)
        assert np.array_equal(got.reshape(-1).view(np.uint32), a["D"].reshape(-1).view(np.uint32))
    for m in means:
        got = oracle.elas_adaptive_mean(m["D_in"], m["width"], m["height"], m["subsampling"])
        assert np.array_equal(got.reshape(-1).view(np.uint32), m["D_out"].reshape(-1).view(np.uint32))
    want = np.load(os.path.join(GOLDEN, "elas_capture.npz"))["D_can"]      # (checked against the reference pipeline when stored)
    assert np.array_equal(oracle.elas_support_candidates(calls[0]), want) and (want >= 0).sum() > 20


# ------------------------------------------------------------------ the HIP path, through the C ABI
def _hip_disparity(e):
    def f(a):
        return e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], a["I1_desc"], a["I2_desc"], a["right_image"],
                                  a["width"], a["height"])
    return f


@pytest.mark.gpu
def test_hip_reproduces_the_committed_capture_of_the_reference():
    from plvs_amd.elas import ElasGPU
    calls, means = golden_capture()
    e = ElasGPU()
    for a in calls:
        got = _hip_disparity(e)(a)
        assert np.array_equal(got.reshape(-1).view(np.uint32), a["D"].reshape(-1).view(np.uint32)), a["right_image"]
    # the second image of a pair on the descriptors the first call staged
    a = dict(calls[1], I1_desc=None, I2_desc=None)
    assert np.array_equal(_hip_disparity(e)(a).reshape(-1).view(np.uint32), calls[1]["D"].reshape(-1).view(np.uint32))
    # the candidate grid of computeSupportMatches, and both images on the pair IT staged
    want = np.load(os.path.join(GOLDEN, "elas_capture.npz"))["D_can"]
    e2 = ElasGPU()
    assert np.array_equal(e2.supportCandidates(calls[0]["I1_desc"], calls[0]["I2_desc"], calls[0]["width"], calls[0]["height"]), want)
    for c in calls:
        got = _hip_disparity(e2)(dict(c, I1_desc=None, I2_desc=None))
        assert np.array_equal(got.reshape(-1).view(np.uint32), c["D"].reshape(-1).view(np.uint32))
    for m in means:
        em = ElasGPU(ElasGPU.Parameters(subsampling=bool(m["subsampling"])))
        got = em.adaptiveMean(m["D_in"], m["width"], m["height"])
        assert np.array_equal(got.reshape(-1).view(np.uint32), m["D_out"].reshape(-1).view(np.uint32)), m["subsampling"]


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300), (333, 201)])
def test_hip_equals_oracle_and_reference_on_the_reference_pipelines_own_arguments(oracle, subsampling, size):
    from plvs_amd.elas import ElasGPU
    left, right = pair(*size)
    disp_calls, mean_calls, _ = elas_ref.capture(left, right, subsampling=subsampling, plvs=False)
    e = ElasGPU(ElasGPU.Parameters(subsampling=subsampling))
    for a in disp_calls:
        got = _hip_disparity(e)(a)
        assert np.array_equal(got.reshape(-1).view(np.uint32), a["D"].view(np.uint32)), f"right_image={a['right_image']}"
        assert np.array_equal(got.view(np.uint32), oracle.elas_compute_disparity(a).view(np.uint32))
    for m in mean_calls:
        got = e.adaptiveMean(m["D_in"], m["width"], m["height"])
        assert np.array_equal(got.reshape(-1).view(np.uint32), m["D_out"].view(np.uint32))


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
def test_reference_pipeline_with_the_hip_path_in_elasgpus_place_equals_the_reference(subsampling):
    """What a PLVS built against this library runs: Elas::process with computeDisparity and adaptiveMean on the device."""
    from plvs_amd.elas import ElasGPU
    left, right = pair()
    want = elas_ref.reference(left, right, subsampling=subsampling, plvs=True)
    e = ElasGPU(ElasGPU.Parameters(subsampling=subsampling))
    got = elas_ref.run_with(left, right, _hip_disparity(e), lambda D, w, h, sub: e.adaptiveMean(D, w, h), subsampling=subsampling,
                            plvs=True)
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    assert (want[0] >= 0).mean() > 0.3


@pytest.mark.gpu
def test_hip_elas_rejects_bad_arguments():
    from plvs_amd import _lib
    from plvs_amd.elas import ElasGPU
    calls, _ = golden_capture()
    a = calls[0]
    e = ElasGPU()
    with pytest.raises(_lib.PlvsHipError):      # no descriptors staged yet
        e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, 0, a["width"], a["height"])
    bad = a["tri"].copy()
    bad["c2"][0] = len(a["support"])
    with pytest.raises(_lib.PlvsHipError):      # a corner that is no support point
        e.computeDisparity(a["support"], bad, a["grid"], a["grid_dims"], a["I1_desc"], a["I2_desc"], 0, a["width"], a["height"])
    # no triangles: nothing is matched
    D = e.computeDisparity(a["support"], a["tri"][:0], a["grid"], a["grid_dims"], a["I1_desc"], a["I2_desc"], 0, a["width"],
                           a["height"])
    assert np.all(D == -10.0)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300), (333, 201)])
def test_hip_candidate_grid_equals_oracle_and_feeds_the_reference_pipeline(oracle, subsampling, size):
    """Elas::computeSupportMatches' candidate loop on the device: equal to the oracle's grid point by point; and the
    reference pipeline with that loop, computeDisparity and adaptiveMean all on the device gives the reference's maps."""
    from plvs_amd.elas import ElasGPU
    left, right = pair(*size)
    e = ElasGPU(ElasGPU.Parameters(subsampling=subsampling))
    grids = []

    def candidates(a):
        got = e.supportCandidates(a["I1_desc"], a["I2_desc"], a["width"], a["height"])
        want = oracle.elas_support_candidates(a)
        assert np.array_equal(got, want), f"{int((got != want).sum())} grid points differ"
        grids.append(got)
        return got

    def disparity(a):        # the pair is staged by supportCandidates
        return e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, a["right_image"], a["width"],
                                  a["height"])
    want = elas_ref.reference(left, right, subsampling=subsampling, plvs=True)
    got = elas_ref.run_with(left, right, disparity, lambda D, w, h, sub: e.adaptiveMean(D, w, h), subsampling=subsampling,
                            plvs=True, support_candidates=candidates)
    assert len(grids) == 1 and (grids[0] >= 0).sum() > 50
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("middlebury", [False, True])
@pytest.mark.parametrize("subsampling", [False, True])
def test_reference_pipeline_with_every_device_stage_equals_the_reference(oracle, subsampling, middlebury):
    """Elas::process with its candidate loop, computeDisparity, leftRightConsistencyCheck, removeSmallSegments,
    gapInterpolation and adaptiveMean on the device (ROBOTICS parameters as PLVS sets them, and the MIDDLEBURY set with
    add_corners and gaps of any width); every post-processing stage also against the oracle on the map it was given."""
    from plvs_amd.elas import ElasGPU
    left, right = pair()
    mode = 2 if middlebury else 0
    prm = dict(lr=2, size=200, sim=1.0, gap=5000 if middlebury else 3, corners=middlebury)
    P = ElasGPU.Parameters(subsampling=subsampling, ipol_gap_width=prm["gap"], add_corners=prm["corners"],
                           **(dict(support_threshold=0.95, gamma=5.0, sradius=3.0, match_texture=0) if middlebury else {}))
    e = ElasGPU(P)
    h, w = left.shape
    want = elas_ref._run(left, right, subsampling, mode, None, None)

    def lr(D1, D2):
        o1, o2 = D1.copy(), D2.copy()
        oracle.elas_left_right_check(o1, o2, subsampling, prm["lr"])
        D1[:], D2[:] = e.leftRightConsistencyCheck(D1, D2, w, h)
        assert np.array_equal(D1.view(np.uint32), o1.view(np.uint32)) and np.array_equal(D2.view(np.uint32), o2.view(np.uint32))

    def seg(D):
        o = D.copy()
        oracle.elas_remove_small_segments(o, subsampling, prm["size"], prm["sim"])
        D[:] = e.removeSmallSegments(D, w, h)
        assert np.array_equal(D.view(np.uint32), o.view(np.uint32)), f"{int((D != o).sum())} pixels differ"

    def gap(D):
        o = D.copy()
        oracle.elas_gap_interpolation(o, subsampling, prm["gap"], prm["corners"])
        D[:] = e.gapInterpolation(D, w, h)
        assert np.array_equal(D.view(np.uint32), o.view(np.uint32))

    def disparity(a):
        return e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, a["right_image"], a["width"],
                                  a["height"])
    got = elas_ref.run_with(left, right, disparity, lambda D, ww, hh, sub: e.adaptiveMean(D, ww, hh), subsampling=subsampling,
                            plvs=mode, support_candidates=lambda a: e.supportCandidates(a["I1_desc"], a["I2_desc"], a["width"], a["height"]),
                            post=dict(left_right_check=lr, remove_small_segments=seg, gap_interpolation=gap))
    for g, wv in zip(got, want):
        assert np.array_equal(g.view(np.uint32), wv.view(np.uint32))
    assert (want[0] >= 0).mean() > 0.3


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("subsampling", [False, True])
@pytest.mark.parametrize("size", [(1241, 376), (640, 300), (333, 201)])
def test_hip_descriptors_from_the_images_and_the_whole_chain_on_them(oracle, subsampling, size):
    """libelas::Descriptor on the device from the two images: the staged descriptor images equal the oracle's (= the
    reference's) byte for byte; and the reference pipeline run with every device stage reading THOSE (the host descriptors
    the pipeline passes are ignored) gives the reference's maps."""
    from plvs_amd.elas import ElasGPU
    left, right = pair(*size)
    h, w = left.shape
    e = ElasGPU(ElasGPU.Parameters(subsampling=subsampling))
    e.setImages(left, right)
    d1, d2 = e.descriptors()
    assert np.array_equal(d1, oracle.elas_descriptor(left, subsampling)) and np.array_equal(d2, oracle.elas_descriptor(right, subsampling))
    want = elas_ref.reference(left, right, subsampling=subsampling, plvs=True)

    def disparity(a):
        return e.computeDisparity(a["support"], a["tri"], a["grid"], a["grid_dims"], None, None, a["right_image"], w, h)

    def lr(D1, D2):
        D1[:], D2[:] = e.leftRightConsistencyCheck(D1, D2, w, h)

    def seg(D):
        D[:] = e.removeSmallSegments(D, w, h)

    def gap(D):
        D[:] = e.gapInterpolation(D, w, h)
    got = elas_ref.run_with(left, right, disparity, lambda D, ww, hh, sub: e.adaptiveMean(D, ww, hh), subsampling=subsampling, plvs=True,
                            support_candidates=lambda a: e.supportCandidates(None, None, w, h),
                            post=dict(left_right_check=lr, remove_small_segments=seg, gap_interpolation=gap))
    for g, wv in zip(got, want):
        assert np.array_equal(g.view(np.uint32), wv.view(np.uint32))
