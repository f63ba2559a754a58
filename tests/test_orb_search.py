"""ORBmatcher::SearchByProjection(Frame, MapPoints) (reference src/ORBmatcher.cc:71-244): the
oracle restatement (oracle/orb_search.c) and the library function built on the batched
candidate-pair Hamming kernel must give identical assignments."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib


def make_case(seed, n=1500, m=900, w=640, h=480, occupied_frac=0.1):
    """A frame of n keypoints and m map points that mostly re-observe them."""
    from plvs_amd.orbmatcher import FrameView, MapPointView
    rng = np.random.default_rng(seed)
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    x = rng.uniform(0, w, n).astype(np.float32)
    y = rng.uniform(0, h, n).astype(np.float32)
    octave = rng.integers(0, 8, n).astype(np.int32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u_right = np.where(rng.random(n) < 0.7, x - rng.uniform(2, 40, n), -1).astype(np.float32)
    F = FrameView(x, y, octave, u_right, desc, 0.0, 0.0, 64.0 / w, 48.0 / h, scale)
    src = rng.integers(0, n, m)
    noise = rng.normal(0, 3.0, (m, 2)).astype(np.float32)
    mdesc = desc[src].copy()
    flips = rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) \
        & rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8)
    mdesc ^= flips                                   # ~16 flipped bits: distances around the thresholds
    rnd = rng.random(m) < 0.15                        # some map points that match nothing
    mdesc[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    level = np.clip(octave[src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32)
    M = MapPointView(track_in_view=(rng.random(m) < 0.9), bad=(rng.random(m) < 0.03),
                     proj_x=x[src] + noise[:, 0], proj_y=y[src] + noise[:, 1],
                     proj_xr=np.where(u_right[src] > 0, u_right[src] + rng.normal(0, 2.0, m), -1),
                     view_cos=rng.uniform(0.99, 1.0, m), track_depth=rng.uniform(0.5, 60.0, m), level=level,
                     desc=mdesc, has_obs=(rng.random(m) < 0.97))
    occupied = (rng.random(n) < occupied_frac).astype(np.uint8)
    return F, M, occupied


def oracle_search(lib, F, M, th, far, th_far, ratio, occupied):
    fc, mc = F.as_c(), M.as_c()
    assigned = np.full(fc.n, -7, np.int32)
    f = lib.oracle_orb_search_by_projection
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                  ctypes.c_void_p, ctypes.c_void_p]
    n = f(ctypes.byref(fc), ctypes.byref(mc), th, int(far), th_far, ratio,
          occupied.ctypes.data_as(ctypes.c_void_p), assigned.ctypes.data_as(ctypes.c_void_p))
    return n, assigned


def test_oracle_search_by_projection_properties(oracle):
    """Known answers on a hand-made frame + invariants on a random one."""
    from plvs_amd.orbmatcher import FrameView, MapPointView
    lib = oracle.lib
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    d = np.zeros((3, 32), np.uint8)
    d[1, 0] = 0xFF            # keypoint 1: 8 bits from keypoint 0
    d[2, :13] = 0xFF          # keypoint 2: 104 bits away: above TH_HIGH
    F = FrameView(np.array([100, 101.5, 300], np.float32), np.array([100, 100, 300], np.float32),
                  np.array([2, 2, 2], np.int32), np.array([-1, -1, -1], np.float32), d, 0, 0, 0.1, 0.1, scale)
    one = lambda v, t=np.float32: np.array([v], t)
    M = MapPointView(one(1, np.uint8), one(0, np.uint8), one(100.2), one(100.1), one(-1), one(0.9999), one(3.0),
                     one(2, np.int32), d[:1].copy())
    # best = kp 0 (distance 0), second = kp 1 (distance 8) on the same level: 0 <= ratio * 8 -> matched
    n, a = oracle_search(lib, F, M, 1.0, False, 0.0, 0.8, np.zeros(3, np.uint8))
    assert n == 1 and list(a) == [0, -1, -1]
    # keypoint 0 occupied: the best free candidate is kp 1 (8 <= 100), no second -> bestLevel2 = -1 != 2 -> matched
    n, a = oracle_search(lib, F, M, 1.0, False, 0.0, 0.8, np.array([1, 0, 0], np.uint8))
    assert n == 1 and list(a) == [-1, 0, -1]
    # a far point is skipped when bFarPoints is set
    n, a = oracle_search(lib, F, M, 1.0, True, 2.0, 0.8, np.zeros(3, np.uint8))
    assert n == 0 and list(a) == [-1, -1, -1]
    # random case: every assignment respects the window, the level band, TH_HIGH and uniqueness
    F, M, occ = make_case(3)
    n, a = oracle_search(lib, F, M, 1.0, False, 0.0, 0.8, occ)
    got = np.nonzero(a >= 0)[0]
    # (a map point without observations does not block its keypoint: a later one may take it over,
    # and the reference counts both)
    assert n >= len(got) > 100 and n - len(got) <= int((~M.has_obs.astype(bool)).sum())
    assert not occ[got].any()
    for i in got:
        k = a[i]
        r = (2.5 if M.view_cos[k] > 0.998 else 4.0) * F.scale_factors[M.level[k]]
        assert abs(F.x[i] - M.proj_x[k]) < r and abs(F.y[i] - M.proj_y[k]) < r
        assert M.level[k] - 1 <= F.octave[i] <= M.level[k]
        assert oracle.descriptor_distance(M.desc[k], F.desc[i]) <= 100
        assert M.track_in_view[k] and not M.bad[k]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,far", [(1, 1.0, False), (2, 3.0, False), (5, 1.0, True), (8, 5.0, True)])
def test_hip_search_by_projection_matches_oracle(oracle, seed, th, far):
    from plvs_amd.orbmatcher import ORBmatcher
    F, M, occ = make_case(seed)
    want_n, want = oracle_search(oracle.lib, F, M, th, far, 40.0, 0.8, occ)
    got_n, got = ORBmatcher(0.8, True).SearchByProjection(F, M, th, far, 40.0, occupied=occ)
    assert got_n == want_n > 50
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_hamming_pairs_and_edge_cases(oracle):
    from plvs_amd import _lib
    from plvs_amd.orbmatcher import FrameView, MapPointView, ORBmatcher
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    pq = rng.integers(0, 50, 4000).astype(np.int32)
    pt = rng.integers(0, 70, 4000).astype(np.int32)
    d = np.zeros(4000, np.int32)
    _lib.check(_lib.lib.plvs_hip_hamming_pairs(_lib.np_ptr(q), 50, _lib.np_ptr(t), 70, _lib.np_ptr(pq), _lib.np_ptr(pt),
                                               4000, _lib.np_ptr(d)))
    ref = np.unpackbits(q[pq] ^ t[pt], axis=1).sum(1)
    assert np.array_equal(d, ref)
    assert ORBmatcher.DescriptorDistance(q[3], t[9]) == oracle.descriptor_distance(q[3], t[9])
    # empty inputs: nothing assigned, zero matches
    F, M, occ = make_case(4, n=40, m=0)
    n, a = ORBmatcher(0.8).SearchByProjection(F, M, occupied=occ)
    assert n == 0 and (a == -1).all()
    # out-of-range pair index is an error, not a fault
    bad = np.array([999], np.int32)
    rc = _lib.lib.plvs_hip_hamming_pairs(_lib.np_ptr(q), 50, _lib.np_ptr(t), 70, _lib.np_ptr(bad), _lib.np_ptr(bad), 1,
                                         _lib.np_ptr(d))
    assert rc != 0


# ------------------------------------------------------------------ frame to frame (M2)
def make_ff_case(seed, n=1500, w=640, h=480):
    from plvs_amd.orbmatcher import LastFrameView
    F, _, occ = make_case(seed, n=n, m=1)
    rng = np.random.default_rng(seed + 100)
    cur_angle = rng.uniform(0, 360, n).astype(np.float32)
    nl = 1200
    src = rng.integers(0, n, nl)
    desc = F.desc[src].copy()
    desc ^= (rng.integers(0, 256, (nl, 32), dtype=np.uint8) & rng.integers(0, 256, (nl, 32), dtype=np.uint8)
             & rng.integers(0, 256, (nl, 32), dtype=np.uint8))
    u = (F.x[src] + rng.normal(0, 4.0, nl)).astype(np.float32)
    v = (F.y[src] + rng.normal(0, 4.0, nl)).astype(np.float32)
    far = rng.random(nl) < 0.05
    u[far] += 2000                                   # projections outside the image bounds
    invz = rng.uniform(-0.05, 1.0, nl).astype(np.float32)   # a few behind the camera
    ang = (cur_angle[src] + 20 + rng.normal(0, 6, nl)).astype(np.float32)
    wild = rng.random(nl) < 0.2
    ang[wild] = rng.uniform(0, 360, int(wild.sum()))
    L = LastFrameView(valid=(rng.random(nl) < 0.8), u=u, v=v, invz=invz,
                      octave=np.clip(F.octave[src] + rng.integers(-1, 2, nl), 0, 7), angle=ang, desc=desc,
                      has_obs=(rng.random(nl) < 0.97))
    return F, cur_angle, float(w), float(h), 40.0, L, occ


def oracle_search_ff(lib, F, cur_angle, max_x, max_y, mbf, L, th, fwd, bwd, check, occ):
    fc, lc = F.as_c(), L.as_c()
    assigned = np.full(fc.n, -7, np.int32)
    f = lib.oracle_orb_search_by_projection_ff
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                  ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = f(ctypes.byref(fc), p(cur_angle), max_x, max_y, mbf, ctypes.byref(lc), th, fwd, bwd, check, p(occ), p(assigned))
    return n, assigned


def test_oracle_search_last_frame_properties(oracle):
    F, ang, mx, my, mbf, L, occ = make_ff_case(2)
    n0, a0 = oracle_search_ff(oracle.lib, F, ang, mx, my, mbf, L, 15.0, 0, 0, 0, occ)
    got = np.nonzero(a0 >= 0)[0]
    assert n0 >= len(got) > 100
    for i2 in got:
        i = a0[i2]
        assert L.valid[i] and L.invz[i] >= 0 and 0 <= L.u[i] <= mx and 0 <= L.v[i] <= my and not occ[i2]
        r = 15.0 * F.scale_factors[L.octave[i]]
        assert abs(F.x[i2] - L.u[i]) < r and abs(F.y[i2] - L.v[i]) < r
        assert L.octave[i] - 1 <= F.octave[i2] <= L.octave[i] + 1
        assert oracle.descriptor_distance(L.desc[i], F.desc[i2]) <= 100
    # the rotation check only removes matches; forward / backward restrict the octave band
    n1, a1 = oracle_search_ff(oracle.lib, F, ang, mx, my, mbf, L, 15.0, 0, 0, 1, occ)
    assert n1 < n0 and set(np.nonzero(a1 >= 0)[0]) <= set(got)
    nf, af = oracle_search_ff(oracle.lib, F, ang, mx, my, mbf, L, 15.0, 1, 0, 0, occ)
    for i2 in np.nonzero(af >= 0)[0]:
        assert F.octave[i2] >= L.octave[af[i2]]
    nb, ab = oracle_search_ff(oracle.lib, F, ang, mx, my, mbf, L, 15.0, 0, 1, 0, occ)
    for i2 in np.nonzero(ab >= 0)[0]:
        assert F.octave[i2] <= L.octave[ab[i2]]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,fwd,bwd,check", [(1, 15.0, 0, 0, 1), (2, 7.0, 0, 0, 1), (3, 15.0, 1, 0, 1),
                                                   (4, 30.0, 0, 1, 0)])
def test_hip_search_last_frame_matches_oracle(oracle, seed, th, fwd, bwd, check):
    from plvs_amd.orbmatcher import ORBmatcher
    F, ang, mx, my, mbf, L, occ = make_ff_case(seed)
    want_n, want = oracle_search_ff(oracle.lib, F, ang, mx, my, mbf, L, th, fwd, bwd, check, occ)
    got_n, got = ORBmatcher(0.9, bool(check)).SearchByProjectionLastFrame(F, ang, mx, my, mbf, L, th, bool(fwd),
                                                                         bool(bwd), occupied=occ)
    assert got_n == want_n > 30
    assert np.array_equal(got, want)


# ----------------------------------------------------------------- SearchByBoW (M4)
def make_bow_case(seed, nk=1800, nf=2000, nodes=90):
    """A key frame and a frame re-observing it: descriptors a few bits apart, the vocabulary node a
    function of the clean descriptor so that most true pairs share a node; some land elsewhere."""
    from plvs_amd.orbmatcher import FeatureVector
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
    node_of = (base[:, 0].astype(np.int64) * 7 + base[:, 1]) % nodes * 3 + 10      # sparse, unordered ids
    kf_desc = base.copy()
    src = rng.integers(0, nk, nf)
    f_desc = base[src].copy()
    flips = rng.integers(0, 256, (nf, 32), dtype=np.uint8) & rng.integers(0, 256, (nf, 32), dtype=np.uint8) \
        & rng.integers(0, 256, (nf, 32), dtype=np.uint8)
    f_desc ^= flips                                                               # ~32 bits: around TH_LOW
    exact = rng.random(nf) < 0.3
    f_desc[exact] = base[src[exact]]                                              # exact copies -> ties at 0
    f_node = node_of[src].copy()
    stray = rng.random(nf) < 0.1
    f_node[stray] = rng.integers(0, nodes, int(stray.sum())) * 3 + 10 + rng.integers(0, 2, int(stray.sum()))
    kf_nodes, f_nodes = {}, {}
    for i in rng.permutation(nk):
        kf_nodes.setdefault(int(node_of[i]), []).append(int(i))
    for i in rng.permutation(nf):
        f_nodes.setdefault(int(f_node[i]), []).append(int(i))
    kf_valid = (rng.random(nk) < 0.8).astype(np.uint8)
    kf_angle = rng.uniform(0, 360, nk).astype(np.float32)
    f_angle = (kf_angle[src] + np.where(rng.random(nf) < 0.8, 25.0, rng.uniform(0, 360, nf))
               + rng.normal(0, 4.0, nf)).astype(np.float32) % np.float32(360.0)
    return FeatureVector(kf_nodes), kf_desc, kf_valid, kf_angle, FeatureVector(f_nodes), f_desc, f_angle


def oracle_search_bow(lib, KV, kd, kv, ka, FV, fd, fa, ratio, check):
    assigned = np.full(fd.shape[0], -7, np.int32)
    f = lib.oracle_orb_search_by_bow
    f.restype = ctypes.c_int
    vp = ctypes.c_void_p
    f.argtypes = [vp, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_float, ctypes.c_int, vp]
    kc, fc = KV.as_c(), FV.as_c()
    p = lambda a: a.ctypes.data_as(vp)
    n = f(ctypes.byref(kc), p(kd), kd.shape[0], p(kv), p(ka), ctypes.byref(fc), p(fd), fd.shape[0], p(fa), ratio,
          check, p(assigned))
    return n, assigned


def test_oracle_search_by_bow_properties(oracle):
    KV, kd, kv, ka, FV, fd, fa = make_bow_case(1)
    n0, a0 = oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, 0.7, 0)
    got = np.nonzero(a0 >= 0)[0]
    assert n0 == len(got) > 300
    node_kf = {int(i): int(KV.node_id[a]) for a in range(len(KV.node_id))
               for i in KV.index[KV.offset[a]:KV.offset[a + 1]]}
    node_f = {int(i): int(FV.node_id[a]) for a in range(len(FV.node_id))
              for i in FV.index[FV.offset[a]:FV.offset[a + 1]]}
    assert len(set(a0[got])) == len(got)                       # a key-frame keypoint is visited once
    for i_f in got:
        k = int(a0[i_f])
        assert kv[k] and node_kf[k] == node_f[int(i_f)]
        assert oracle.descriptor_distance(kd[k], fd[i_f]) <= 50
    # a stricter ratio or the rotation check only removes matches
    n1, a1 = oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, 0.5, 0)
    assert n1 < n0
    n2, a2 = oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, 0.7, 1)
    assert 0 < n2 < n0 and set(np.nonzero(a2 >= 0)[0]) <= set(got)
    # no common node -> nothing
    from plvs_amd.orbmatcher import FeatureVector
    n3, a3 = oracle_search_bow(oracle.lib, FeatureVector({1: [0, 1]}), kd, kv, ka, FeatureVector({2: [0, 1]}), fd, fa,
                               0.7, 1)
    assert n3 == 0 and (a3 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,check", [(1, 0.7, 1), (2, 0.7, 0), (3, 0.9, 1), (4, 0.5, 1)])
def test_hip_search_by_bow_matches_oracle(oracle, seed, ratio, check):
    from plvs_amd import _lib
    from plvs_amd.orbmatcher import FeatureVector, ORBmatcher
    KV, kd, kv, ka, FV, fd, fa = make_bow_case(seed)
    want_n, want = oracle_search_bow(oracle.lib, KV, kd, kv, ka, FV, fd, fa, ratio, check)
    got_n, got = ORBmatcher(ratio, bool(check)).SearchByBoW(KV, kd, kv, ka, FV, fd, fa)
    assert got_n == want_n > 100
    assert np.array_equal(got, want)
    # empty sides, disjoint vocabularies, malformed vectors
    m = ORBmatcher(ratio, bool(check))
    n, a = m.SearchByBoW(KV, kd, kv, ka, FeatureVector({}), fd, fa)
    assert n == 0 and (a == -1).all()
    n, a = m.SearchByBoW(FeatureVector({5: [0]}), kd, kv, ka, FeatureVector({6: [0]}), fd, fa)
    assert n == 0 and (a == -1).all()
    with pytest.raises(_lib.PlvsHipError):
        m.SearchByBoW(FeatureVector({5: [kd.shape[0]]}), kd, kv, ka, FV, fd, fa)      # index out of range
