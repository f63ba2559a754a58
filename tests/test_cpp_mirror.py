"""include/plvs_hip.hpp — the C++ host-side mirror of the PLVS interfaces (ORBextractor, LineExtractor,
BinaryDescriptorMatcher, ComputeStereoMatches, PointCloudGenerator, PointCloudMapChisel / Voxblox): it must
compile against nothing but the C ABI, and a C++ program using it must produce byte for byte what the Python
mirror produces through the same library (the Python path is what the parity tests pin against the oracle)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "mirror_smoke.cpp")


def build(out):
    lib_dir = os.path.join(ROOT, "plvs_amd", "lib")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                    "-L", lib_dir, "-l:libplvs_hip.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out],
                   check=True)


def test_cpp_mirror_compiles_against_the_c_abi_only(tmp_path):
    build(str(tmp_path / "mirror_smoke"))


@pytest.mark.gpu
def test_cpp_mirror_matches_python_mirror(tmp_path):
    from plvs_amd import cloudgen
    from plvs_amd.lines import LineExtractor
    from plvs_amd.matcher import BinaryDescriptorMatcher
    from plvs_amd.orb import ORBextractor
    from plvs_amd.orbmatcher import ORBmatcher
    from plvs_amd.stereo import StereoMatcher
    from plvs_amd.tsdf import PointCloudMapChisel
    from tests.oracle_lib import golden
    exe = str(tmp_path / "mirror_smoke")
    build(exe)
    out = tmp_path / "out"
    out.mkdir()
    gold = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([exe, os.path.join(gold, "urban1_1241x376.pgm"), os.path.join(gold, "urban1_right_1241x376.pgm"),
                        str(out)], check=True, capture_output=True, text=True)
    lines_out = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l and not l.startswith("PointCloudMap")}
    load = lambda name, dt: np.fromfile(str(out / (name + ".bin")), dtype=dt)

    left, right = golden("urban1_1241x376.pgm"), golden("urban1_right_1241x376.pgm")
    exl, exr = ORBextractor(2000, 1.2, 8, 20, 7), ORBextractor(2000, 1.2, 8, 20, 7)
    mono, kl, dl = exl(left)
    _, kr, dr = exr(right)
    assert int(lines_out["orb_left"][0]) == mono and int(lines_out["orb_left"][1]) == len(kl)
    assert load("orb_keys", np.uint8).tobytes() == kl.tobytes() and load("orb_desc", np.uint8).tobytes() == dl.tobytes()
    assert lines_out["orb_empty"] == ["-1", "0"]                       # empty image: -1, no keypoints
    mb, mbf = np.float32(386.1448) / np.float32(718.856), np.float32(386.1448)
    u, z = StereoMatcher(exl, exr).ComputeStereoMatches(kl, dl, kr, dr, mb, mbf)
    assert load("stereo_uright", np.float32).tobytes() == u.tobytes()
    assert load("stereo_depth", np.float32).tobytes() == z.tobytes()

    lx, lx2 = LineExtractor(100), LineExtractor(100)
    ll, ldl = lx(left)
    lr, ldr = lx2(right)
    assert int(lines_out["lines"][0]) == len(ll)
    assert load("lines_keys", np.uint8).tobytes() == ll.tobytes() and load("lines_desc", np.uint8).tobytes() == ldl.tobytes()
    from plvs_amd.lines import LSDOptions

    class LsdExtractor(LineExtractor):
        skUseLsdExtractor = True
    lsd_kl, lsd_kd = LsdExtractor(100, LSDOptions(numOctaves=3, scale=float(np.float32(1.2)), min_length=0.025, refine=1, log_eps=1.0,
                                          density_th=0.6))(left)
    assert int(lines_out["lsd_lines"][0]) == len(lsd_kl) > 50
    assert load("lsd_keys", np.uint8).tobytes() == lsd_kl.tobytes() and load("lsd_desc", np.uint8).tobytes() == lsd_kd.tobytes()
    matches = []
    BinaryDescriptorMatcher().knnMatch(ldl, ldr, matches, k=2)
    flat = np.array([[m.queryIdx, m.trainIdx, int(m.distance)] for row in matches for m in row], np.int32).reshape(-1)
    assert int(lines_out["knn"][0]) == len(matches) and np.array_equal(load("knn", np.int32), flat)
    assert int(lines_out["knn"][2]) == ORBmatcher.DescriptorDistance(dl[0], dl[1])

    # the search functions through both mirrors
    from plvs_amd.linematcher import LineMatcher
    from plvs_amd.orbmatcher import FeatureVector
    valid = np.ones(len(ll), np.uint8)
    valid[3::7] = 0
    lm = LineMatcher(0.8, True)
    n1, a1 = lm.SearchByKnnLastFrame(ldl, valid, ll["angle"], ldr, lr["angle"])
    n2, a2 = lm.SearchByKnn(ldl, valid, ll["angle"], ldr, lr["angle"])
    n3, sm, sv = lm.SearchStereoMatchesByKnn(ldl, ll["angle"], ll["octave"], ldr, lr["angle"], lr["octave"])
    assert [int(x) for x in lines_out["line_search"]] == [n1, n2, n3, len(sm)] and n1 > 5
    assert np.array_equal(load("line_ff", np.int32), a1) and np.array_equal(load("line_kf", np.int32), a2)
    sflat = np.stack([sm["queryIdx"], sm["trainIdx"], sm["distance"].astype(np.int32), sv.astype(np.int32)], -1).reshape(-1)
    assert np.array_equal(load("line_stereo", np.int32), sflat)

    def featvec(desc):
        nodes = {}
        for i, b in enumerate(desc[:, 0] >> 2):
            nodes.setdefault(int(b), []).append(i)
        return FeatureVector(nodes)
    nb, ab = ORBmatcher(0.7, True).SearchByBoW(featvec(dl), dl, np.ones(len(kl), np.uint8), kl["angle"], featvec(dr), dr,
                                               kr["angle"])
    assert int(lines_out["bow"][0]) == nb > 50 and np.array_equal(load("bow", np.int32), ab)

    from plvs_amd.sgm import StereoSGM
    disp = StereoSGM(1240, 376).execute(left[:, :1240], right[:, :1240])
    assert int(lines_out["sgm"][0]) == int((disp > 0).sum()) and load("sgm", np.uint8).tobytes() == disp.tobytes()

    W, H = 320, 240
    fx, fy, cx, cy = 258.65, 258.23, 159.3, 127.6
    depth = load("depth_img", np.float32).reshape(H, W)
    color = load("color_img", np.uint8).reshape(H, W, 3)
    gen = cloudgen.PointCloudGenerator(W, H, cloudgen.InitCamGridPoints(W, H, 2, fx, fy, cx, cy), step=2, min_depth=0.1,
                                       max_depth=5.0)
    cloud, p2p = gen.GeneratePointCloudInCameraFrameBGRA(color, depth, 7)
    assert load("cloud", np.uint8).tobytes() == cloud.tobytes() and np.array_equal(load("p2p", np.int32).reshape(H, W), p2p)

    Twc = np.array([[1, 0, 0, 0.1], [0, 1, 0, -0.2], [0, 0, 1, 0.05]], np.float32)
    pc = dict(xyz=np.stack([cloud["x"], cloud["y"], cloud["z"]], -1), rgb=np.stack([cloud["r"], cloud["g"], cloud["b"]], -1),
              kfid=cloud["kfid"])
    m = PointCloudMapChisel(0.05, use_carving=True)
    m.InsertCloudWithDepth(pc, Twc, depth, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy))
    Twc2 = Twc.copy()
    Twc2[0, 3] += np.float32(0.03)
    m.InsertCloudWithDepth(pc, Twc2, depth, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy))
    mc = m.UpdateMap()
    assert int(lines_out["chisel"][0]) == len(mc) > 1000 and int(lines_out["chisel"][1]) == len(m.all_meshes)
    assert load("map_cloud", np.uint8).tobytes() == mc.tobytes()
    again = PointCloudMapChisel(0.05)
    lc = again.LoadMap(mc)
    assert int(lines_out["loadmap"][0]) == len(lc) > 500 and int(lines_out["loadmap"][1]) == len(again.all_meshes)
    assert load("loaded_cloud", np.uint8).tobytes() == lc.tobytes()
    from plvs_amd.tsdf import PointCloudMapVoxblox
    v = PointCloudMapVoxblox(0.05, integration_method="simple")
    vc = dict(xyz=pc["xyz"], rgba=np.stack([cloud["r"], cloud["g"], cloud["b"], cloud["a"]], -1))
    v.InsertCloud(vc, Twc2)
    Twc3 = Twc2.copy()
    Twc3[0, 3] -= np.float32(0.03)
    v.InsertCloud(vc, Twc3)
    vcloud = v.UpdateMap()
    assert int(lines_out["voxblox"][0]) == v.tsdf.num_chunks()
    assert int(lines_out["voxblox"][1]) == len(vcloud) > 1000 and int(lines_out["voxblox"][2]) == len(v.mesh_layer)
    assert load("vmap_cloud", np.uint8).tobytes() == vcloud.tobytes()
    assert PointCloudMapVoxblox.skIntegrationMethod == "fast"     # src/PointCloudMapVoxblox.cc:44
    vf = PointCloudMapVoxblox(0.05)                                # the default IS the reference's default
    vf.InsertCloud(vc, Twc3)
    Twc4 = Twc3.copy()
    Twc4[0, 3] += np.float32(0.03)
    vf.InsertCloud(vc, Twc4)
    fcloud = vf.UpdateMap()
    assert int(lines_out["voxblox_fast"][0]) == vf.tsdf.num_chunks() and int(lines_out["voxblox_fast"][1]) == len(fcloud) > 500
    assert load("vfast_cloud", np.uint8).tobytes() == fcloud.tobytes()
    assert len(fcloud) != len(vcloud)
    # the layer saved and loaded into an empty map meshes to the same cloud
    assert lines_out["voxblox_layer"] == [lines_out["voxblox"][0], lines_out["voxblox"][1], "1"]
    # OnMapChange with cloud deformation, C++ mirror against the python mirror (both over the C ABI)
    dm = PointCloudMapChisel(0.05, bResetOnSparseMapChange=False, bCloudDeformationOnSparseMapChange=True)
    dm.InsertCloud(pc, Twc)
    dm.UpdateMap()
    R = np.array([[0.9998, -0.02, 0.0], [0.02, 0.9998, 0.0], [0.0, 0.0, 1.0]], np.float32)
    dc = dm.OnMapChange({7: (R, np.array([0.04, -0.03, 0.11], np.float32))})
    assert int(lines_out["deform"][0]) == len(dc) > 1000
    assert load("deformed_cloud", np.uint8).tobytes() == dc.tobytes()
    dm.InsertCloud(pc, Twc)
    dc2 = dm.UpdateMap()
    assert int(lines_out["deform"][1]) == len(dc2) > 1000
    assert load("deformed_cloud2", np.uint8).tobytes() == dc2.tobytes()
    assert lines_out["cleared"] == ["0"]
