"""LineMatcher::SearchByProjection (F, LastF) and (F, MapLines) — the guided line searches Tracking runs on
every frame (reference src/LineMatcher.cc:837-1230, 1286-1560; grid lookup src/Frame.cc:1326-1495): oracle
restatement vs the library function (host windows + gates, one batched device Hamming launch, the
reference's greedy loop)."""
import ctypes

import numpy as np
import pytest

from tests.oracle_lib import KEYLINE_DTYPE

W, H = 640, 480
MAX_DIAG = float(np.float32(np.sqrt(np.float32(W * W + H * H))))
SCALE = (1.2 ** np.arange(3)).astype(np.float32)          # Line.scaleFactor 1.2, 3 levels
INV_SIGMA2 = (1.0 / (SCALE * SCALE)).astype(np.float32)


def make_case(seed, n_cur=120, n_last=100, stereo=False, theta_edge=False):
    """A current frame of random segments and a last frame whose map lines project near a subset of them
    (some displaced, some with foreign descriptors, some invalid); with theta_edge the segments are nearly
    horizontal / vertical so that the (theta, d) windows wrap at +-pi/2."""
    rng = np.random.default_rng(seed)
    kl = np.zeros(n_cur, KEYLINE_DTYPE)
    cx, cy = rng.uniform(40, W - 40, n_cur), rng.uniform(40, H - 40, n_cur)
    ang = rng.uniform(-np.pi, np.pi, n_cur)
    if theta_edge:
        ang = rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi], n_cur) + rng.normal(0, 0.03, n_cur)
    ln = rng.uniform(20, 120, n_cur)
    kl["startPointX"], kl["startPointY"] = cx - 0.5 * ln * np.cos(ang), cy - 0.5 * ln * np.sin(ang)
    kl["endPointX"], kl["endPointY"] = cx + 0.5 * ln * np.cos(ang), cy + 0.5 * ln * np.sin(ang)
    kl["angle"] = np.arctan2(kl["endPointY"] - kl["startPointY"], kl["endPointX"] - kl["startPointX"])
    kl["octave"] = rng.integers(0, 3, n_cur)
    kl["lineLength"] = ln
    desc = rng.integers(0, 256, (n_cur, 32), dtype=np.uint8)
    src = rng.integers(0, n_cur, n_last)
    proj = np.zeros((n_last, 6), np.float32)
    jit = rng.normal(0, 0.8, (n_last, 4))
    far = rng.random(n_last) < 0.15
    jit[far] += rng.normal(0, 25, (int(far.sum()), 4))
    proj[:, 0], proj[:, 1] = kl["startPointX"][src] + jit[:, 0], kl["startPointY"][src] + jit[:, 1]
    proj[:, 2], proj[:, 3] = kl["endPointX"][src] + jit[:, 2], kl["endPointY"][src] + jit[:, 3]
    depth = rng.uniform(1.0, 6.0, (n_last, 2))
    proj[:, 4:6] = 1.0 / depth
    ldesc = desc[src].copy()
    ldesc ^= (rng.integers(0, 256, (n_last, 32), dtype=np.uint8) & rng.integers(0, 256, (n_last, 32), dtype=np.uint8)
              & rng.integers(0, 256, (n_last, 32), dtype=np.uint8))
    noise = rng.random(n_last) < 0.2
    ldesc[noise] = rng.integers(0, 256, (int(noise.sum()), 32), dtype=np.uint8)
    octave = np.clip(kl["octave"][src] + rng.integers(-1, 2, n_last), 0, 2).astype(np.int32)
    angle = (kl["angle"][src] + 0.04 + rng.normal(0, 0.03, n_last)).astype(np.float32)
    wild = rng.random(n_last) < 0.1
    angle[wild] = rng.uniform(-np.pi, np.pi, int(wild.sum()))
    valid = (rng.random(n_last) < 0.9).astype(np.uint8)
    has_obs = (rng.random(n_last) < 0.8).astype(np.uint8)
    occupied = (rng.random(n_cur) < 0.1).astype(np.uint8)
    urs = ure = None
    bf = 0.0
    if stereo:
        bf = 40.0
        urs = (kl["startPointX"] - bf / rng.uniform(1.0, 6.0, n_cur)).astype(np.float32)
        ure = (kl["endPointX"] - bf / rng.uniform(1.0, 6.0, n_cur)).astype(np.float32)
        none = rng.random(n_cur) < 0.4
        urs[none] = -1
        ure[none] = -1
        # give the sources of half the map lines a consistent right observation
        for i in range(0, n_last, 2):
            j = src[i]
            urs[j] = kl["startPointX"][j] - bf * proj[i, 4]
            ure[j] = kl["endPointX"][j] - bf * proj[i, 5]
    # the map-line search takes the DEPTHS of the end points (mTrackStartDepth / EndDepth: it divides mbf by them,
    # src/LineMatcher.cc:1419-1423), the frame-to-frame search the projection's inverse depths (:1011-1013)
    proj_map = proj.copy()
    proj_map[:, 4:6] = depth
    return dict(kl=kl, desc=desc, urs=urs, ure=ure, bf=bf, occupied=occupied, valid=valid, proj=proj, proj_map=proj_map,
                octave=octave, angle=angle, ldesc=ldesc, has_obs=has_obs, src=src)


def _p(a):
    return None if a is None else np.ascontiguousarray(a).ctypes.data_as(ctypes.c_void_p)


def oracle_ff(oracle, c, larger, direction, ratio, check):
    f = oracle.lib.oracle_lines_search_by_projection_ff
    f.restype = ctypes.c_int
    vp, i, fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    f.argtypes = [i, vp, vp, vp, vp, fl, vp, vp, fl, vp, i, vp, vp, vp, vp, vp, vp, i, i, fl, i, vp]
    assigned = np.full(len(c["kl"]), -7, np.int32)
    n = f(len(c["kl"]), _p(c["kl"]), _p(c["desc"]), _p(c["urs"]), _p(c["ure"]), c["bf"], _p(SCALE), _p(INV_SIGMA2), MAX_DIAG,
          _p(c["occupied"]), len(c["valid"]), _p(c["valid"]), _p(c["proj"]), _p(c["octave"]), _p(c["angle"]), _p(c["ldesc"]),
          _p(c["has_obs"]), int(larger), int(direction), ratio, int(check), _p(assigned))
    return n, assigned


def oracle_map(oracle, c, larger, ratio):
    f = oracle.lib.oracle_lines_search_by_projection_map
    f.restype = ctypes.c_int
    vp, i, fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    f.argtypes = [i, vp, vp, vp, vp, fl, vp, vp, fl, vp, i, vp, vp, vp, vp, vp, i, fl, vp]
    assigned = np.full(len(c["kl"]), -7, np.int32)
    n = f(len(c["kl"]), _p(c["kl"]), _p(c["desc"]), _p(c["urs"]), _p(c["ure"]), c["bf"], _p(SCALE), _p(INV_SIGMA2), MAX_DIAG,
          _p(c["occupied"]), len(c["valid"]), _p(c["valid"]), _p(c["proj_map"]), _p(c["octave"]), _p(c["ldesc"]), _p(c["has_obs"]),
          int(larger), ratio, _p(assigned))
    return n, assigned


def test_oracle_projection_search_properties(oracle):
    c = make_case(3)
    # (a current line given to a map line WITHOUT observations stays free and can be taken again: counted twice)
    n_any, a_any = oracle_ff(oracle, c, False, 0, 0.8, False)
    assert n_any >= int((a_any >= 0).sum()) > 25
    c["has_obs"] = np.ones_like(c["has_obs"])
    n, a = oracle_ff(oracle, c, False, 0, 0.8, False)
    got = np.nonzero(a >= 0)[0]
    assert n == len(got) > 25
    for i2 in got:
        i = a[i2]
        assert c["valid"][i] and not c["occupied"][i2]
        assert oracle.descriptor_distance(c["ldesc"][i], c["desc"][i2]) <= 110
        assert abs(int(c["kl"]["octave"][i2]) - int(c["octave"][i])) <= 1        # the level window of direction 0
        # both end points of the frame line lie within the chi-square gate of the projected line
        nx, ny = c["proj"][i, 3] - c["proj"][i, 1], c["proj"][i, 0] - c["proj"][i, 2]
        nrm = np.hypot(nx, ny)
        d = (nx * c["proj"][i, 2] + ny * c["proj"][i, 3]) / nrm
        for x, y in ((c["kl"]["startPointX"][i2], c["kl"]["startPointY"][i2]), (c["kl"]["endPointX"][i2], c["kl"]["endPointY"][i2])):
            e = (nx * x + ny * y) / nrm - d
            assert e * e * INV_SIGMA2[c["octave"][i]] <= 3.84 * 1.001
    # most undisturbed projections find their source line
    hit = sum(1 for i2 in got if c["src"][a[i2]] == i2)
    assert hit > 0.8 * len(got)
    # the orientation check only removes matches; a larger search only adds candidates
    n1, a1 = oracle_ff(oracle, c, False, 0, 0.8, True)
    assert n1 <= n and set(np.nonzero(a1 >= 0)[0]) <= set(got)
    # forward / backward restrict the octaves
    _, af = oracle_ff(oracle, c, False, 1, 0.8, False)
    for i2 in np.nonzero(af >= 0)[0]:
        assert c["kl"]["octave"][i2] >= c["octave"][af[i2]]
    _, ab = oracle_ff(oracle, c, False, 2, 0.8, False)
    for i2 in np.nonzero(ab >= 0)[0]:
        assert c["kl"]["octave"][i2] <= c["octave"][ab[i2]]
    # nothing valid -> nothing matched
    c0 = dict(c, valid=np.zeros_like(c["valid"]))
    assert oracle_ff(oracle, c0, False, 0, 0.8, True)[0] == 0
    # the map-line search: levels [l - 1, l], ratio only inside one level
    nm, am = oracle_map(oracle, c, False, 0.8)
    gm = np.nonzero(am >= 0)[0]
    assert nm == len(gm) > 15
    for i2 in gm:
        lv = int(c["octave"][am[i2]])
        assert lv - 1 <= int(c["kl"]["octave"][i2]) <= lv


@pytest.mark.gpu
@pytest.mark.parametrize("seed,stereo,edge", [(1, False, False), (2, True, False), (3, False, True), (4, True, True)])
@pytest.mark.parametrize("larger,direction,ratio,check", [(False, 0, 0.8, True), (True, 1, 0.7, False), (False, 2, 0.9, True)])
def test_hip_projection_search_matches_oracle(oracle, seed, stereo, edge, larger, direction, ratio, check):
    from plvs_amd.linematcher import LineMatcher, line_frame_view
    c = make_case(seed, n_cur=150, n_last=130, stereo=stereo, theta_edge=edge)
    view = line_frame_view(c["kl"], c["desc"], SCALE, INV_SIGMA2, MAX_DIAG, c["urs"], c["ure"], c["bf"])
    m = LineMatcher(ratio, check)
    want_n, want = oracle_ff(oracle, c, larger, direction, ratio, check)
    got_n, got = m.SearchByProjectionLastFrame(view, c["valid"], c["proj"], c["octave"], c["angle"], c["ldesc"],
                                               occupied=c["occupied"], has_obs=c["has_obs"], bLargerSearch=larger,
                                               direction=direction)
    assert want_n >= 0 and got_n == want_n and np.array_equal(got, want)
    want_n, want = oracle_map(oracle, c, larger, ratio)
    got_n, got = m.SearchByProjection(view, c["valid"], c["proj_map"], c["octave"], c["ldesc"], occupied=c["occupied"],
                                      has_obs=c["has_obs"], bLargerSearch=larger)
    assert want_n > 5 and got_n == want_n and np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_projection_search_edge_cases(oracle):
    from plvs_amd import _lib
    from plvs_amd.linematcher import LineMatcher, line_frame_view
    c = make_case(9, n_cur=40, n_last=30)
    m = LineMatcher(0.8, True)
    # an empty current frame / an empty last frame
    empty = line_frame_view(c["kl"][:0], c["desc"][:0], SCALE, INV_SIGMA2, MAX_DIAG)
    assert m.SearchByProjectionLastFrame(empty, c["valid"], c["proj"], c["octave"], c["angle"], c["ldesc"])[0] == 0
    view = line_frame_view(c["kl"], c["desc"], SCALE, INV_SIGMA2, MAX_DIAG)
    n, a = m.SearchByProjectionLastFrame(view, c["valid"][:0], c["proj"][:0], c["octave"][:0], c["angle"][:0], c["ldesc"][:0])
    assert n == 0 and (a == -1).all()
    # a level outside the frame's tables is refused, not read out of bounds
    bad = c["octave"].copy()
    bad[0] = 7
    with pytest.raises(_lib.PlvsHipError):
        m.SearchByProjectionLastFrame(view, np.ones_like(c["valid"]), c["proj"], bad, c["angle"], c["ldesc"])
