"""LineMatcher::SearchByKnn(CurrentFrame, LastFrame) (reference src/LineMatcher.cc:303-447):
oracle restatement vs the library function (device k-NN + host bookkeeping)."""
import ctypes

import numpy as np
import pytest


def make_case(seed, n_last=100, n_cur=110, rot=0.05):
    rng = np.random.default_rng(seed)
    cur = rng.integers(0, 256, (n_cur, 32), dtype=np.uint8)
    src = rng.integers(0, n_cur, n_last)
    last = cur[src].copy()
    last ^= (rng.integers(0, 256, (n_last, 32), dtype=np.uint8) & rng.integers(0, 256, (n_last, 32), dtype=np.uint8)
             & rng.integers(0, 256, (n_last, 32), dtype=np.uint8))          # ~32 flipped bits
    noise = rng.random(n_last) < 0.2
    last[noise] = rng.integers(0, 256, (int(noise.sum()), 32), dtype=np.uint8)
    ang_cur = rng.uniform(-np.pi, np.pi, n_cur).astype(np.float32)
    ang_last = (ang_cur[src] + rot + rng.normal(0, 0.05, n_last)).astype(np.float32)
    wild = rng.random(n_last) < 0.15
    ang_last[wild] = rng.uniform(-np.pi, np.pi, int(wild.sum()))
    valid = (rng.random(n_last) < 0.85).astype(np.uint8)
    return last, valid, ang_last, cur, ang_cur


def run(f, case, ratio, check):
    last, valid, ang_last, cur, ang_cur = case
    assigned = np.full(cur.shape[0], -7, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    return f(p(last), last.shape[0], p(valid), p(ang_last), p(cur), cur.shape[0], p(ang_cur), ratio, int(check),
             p(assigned)), assigned


def oracle_fn(oracle):
    f = oracle.lib.oracle_lines_search_by_knn
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    return f


def test_oracle_line_search_properties(oracle):
    f = oracle_fn(oracle)
    case = make_case(1)
    last, valid, ang_last, cur, ang_cur = case
    n0, a0 = run(f, case, 0.8, False)
    got = np.nonzero(a0 >= 0)[0]
    assert n0 == len(got) > 30                      # one query per matched train line
    for t in got:
        q = a0[t]
        assert valid[q] and oracle.descriptor_distance(last[q], cur[t]) < 110
    # the orientation check only removes matches, and keeps at most three rotation bins
    n1, a1 = run(f, case, 0.8, True)
    kept = np.nonzero(a1 >= 0)[0]
    assert n1 == len(kept) <= n0 and set(kept) <= set(got)
    rot = ang_last[a1[kept]] - ang_cur[kept]
    rot = np.where(rot < 0, rot + 2 * np.pi, rot)
    bins = np.round(rot * (12 / (2 * np.pi))).astype(int) % 12
    assert len(set(bins)) <= 3
    # nothing valid in the last frame -> 0
    n2, a2 = run(f, (last, np.zeros_like(valid), ang_last, cur, ang_cur), 0.8, True)
    assert n2 == 0 and (a2 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,check", [(1, 0.8, True), (2, 0.7, True), (3, 0.9, False), (4, 0.8, True)])
def test_hip_line_search_matches_oracle(oracle, seed, ratio, check):
    from plvs_amd import _lib
    case = make_case(seed, n_last=90 + 7 * seed, n_cur=100 + 3 * seed, rot=0.3 * seed)
    want_n, want = run(oracle_fn(oracle), case, ratio, check)
    last, valid, ang_last, cur, ang_cur = case
    assigned = np.full(cur.shape[0], -7, np.int32)
    n = ctypes.c_int()
    f = _lib.lib.plvs_hip_lines_search_by_knn
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    _lib.check(f(_lib.np_ptr(last), last.shape[0], _lib.np_ptr(valid), _lib.np_ptr(ang_last), _lib.np_ptr(cur),
                 cur.shape[0], _lib.np_ptr(ang_cur), ratio, int(check), _lib.np_ptr(assigned), ctypes.byref(n)))
    assert n.value == want_n > 10
    assert np.array_equal(assigned, want)
    # a single current line: the k-NN has no second neighbour, the ratio test is skipped
    one = (last, valid, ang_last, cur[:1].copy(), ang_cur[:1].copy())
    want_n1, want1 = run(oracle_fn(oracle), one, ratio, check)
    a1 = np.full(1, -7, np.int32)
    _lib.check(f(_lib.np_ptr(last), last.shape[0], _lib.np_ptr(valid), _lib.np_ptr(ang_last), _lib.np_ptr(one[3]), 1,
                 _lib.np_ptr(one[4]), ratio, int(check), _lib.np_ptr(a1), ctypes.byref(n)))
    assert n.value == want_n1 and np.array_equal(a1, want1)


# ------------------------------------------------------------------ key frame -> frame (:156-301)
def oracle_kf_fn(oracle):
    f = oracle.lib.oracle_lines_search_by_knn_kf
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    return f


def test_oracle_kf_line_search_properties(oracle):
    case = make_case(5)
    last, valid, ang_last, cur, ang_cur = case
    n_hi, a_hi = run(oracle_fn(oracle), case, 0.8, False)            # < TH_HIGH
    n_lo, a_lo = run(oracle_kf_fn(oracle), case, 0.8, False)         # <= TH_LOW: a subset
    got = np.nonzero(a_lo >= 0)[0]
    assert 10 < n_lo == len(got) <= n_hi
    for t in got:
        assert valid[a_lo[t]] and oracle.descriptor_distance(last[a_lo[t]], cur[t]) <= 60
        assert a_hi[t] == a_lo[t]
    n1, a1 = run(oracle_kf_fn(oracle), case, 0.8, True)
    assert n1 <= n_lo and set(np.nonzero(a1 >= 0)[0]) <= set(got)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,check", [(1, 0.8, True), (2, 0.7, True), (3, 0.9, False)])
def test_hip_kf_line_search_matches_oracle(oracle, seed, ratio, check):
    from plvs_amd.linematcher import LineMatcher
    case = make_case(10 + seed, n_last=120 + 7 * seed, n_cur=100 + 3 * seed, rot=0.4 * seed)
    want_n, want = run(oracle_kf_fn(oracle), case, ratio, check)
    kf, valid, ang_kf, fr, ang_fr = case
    got_n, got = LineMatcher(ratio, check).SearchByKnn(kf, valid, ang_kf, fr, ang_fr)
    assert got_n == want_n > 10 and np.array_equal(got, want)
    # and the last-frame variant through the same mirror
    want_n, want = run(oracle_fn(oracle), case, ratio, check)
    got_n, got = LineMatcher(ratio, check).SearchByKnnLastFrame(kf, valid, ang_kf, fr, ang_fr)
    assert got_n == want_n and np.array_equal(got, want)
    n0, a0 = LineMatcher(ratio, check).SearchByKnn(kf, np.zeros_like(valid), ang_kf, fr, ang_fr)
    assert n0 == 0 and (a0 == -1).all()


# ------------------------------------------------------------------ stereo (:454-586)
def make_stereo_case(seed, n_left=110, n_right=100):
    rng = np.random.default_rng(seed)
    right = rng.integers(0, 256, (n_right, 32), dtype=np.uint8)
    src = rng.integers(0, n_right, n_left)                        # several left lines per right line
    left = right[src].copy()
    left ^= (rng.integers(0, 256, (n_left, 32), dtype=np.uint8) & rng.integers(0, 256, (n_left, 32), dtype=np.uint8)
             & rng.integers(0, 256, (n_left, 32), dtype=np.uint8) & rng.integers(0, 256, (n_left, 32), dtype=np.uint8))
    noise = rng.random(n_left) < 0.15
    left[noise] = rng.integers(0, 256, (int(noise.sum()), 32), dtype=np.uint8)
    ang_r = rng.uniform(-np.pi, np.pi, n_right).astype(np.float32)
    ang_l = (ang_r[src] + rng.normal(0, 0.03, n_left)).astype(np.float32)
    wild = rng.random(n_left) < 0.2
    ang_l[wild] = rng.uniform(-np.pi, np.pi, int(wild.sum()))
    oct_r = rng.integers(0, 3, n_right).astype(np.int32)
    oct_l = oct_r[src].copy()
    other = rng.random(n_left) < 0.15
    oct_l[other] = (oct_l[other] + 1) % 3
    return left, ang_l, oct_l, right, ang_r, oct_r


def oracle_stereo(oracle, case, ratio, check, dd):
    left, ang_l, oct_l, right, ang_r, oct_r = case
    cap = max(right.shape[0], 1)
    mq, mt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    md, mv = np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
    n_out = ctypes.c_int()
    f = oracle.lib.oracle_lines_search_stereo_by_knn
    f.restype = ctypes.c_int
    vp = ctypes.c_void_p
    f.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                  vp, vp, vp, vp, vp]
    p = lambda a: a.ctypes.data_as(vp)
    n = f(p(left), left.shape[0], p(ang_l), p(oct_l), p(right), right.shape[0], p(ang_r), p(oct_r), ratio, int(check),
          dd, p(mq), p(mt), p(md), p(mv), ctypes.byref(n_out))
    k = n_out.value
    return n, mq[:k], mt[:k], md[:k], mv[:k].astype(bool)


def test_oracle_stereo_line_search_properties(oracle):
    case = make_stereo_case(1)
    left, ang_l, oct_l, right, ang_r, oct_r = case
    n, mq, mt, md, mv = oracle_stereo(oracle, case, 0.8, True, 50)
    assert 10 < n == int(mv.sum()) <= len(mq)
    assert len(set(mt)) == len(mt)                                 # one slot per right line
    for q, t, d in zip(mq, mt, md):
        assert oct_l[q] == oct_r[t] and d < 50 and d == oracle.descriptor_distance(left[q], right[t])
    # the slot of a right line holds the closest of the left lines that chose it
    n0, mq0, mt0, md0, mv0 = oracle_stereo(oracle, case, 0.8, False, 50)
    assert mv0.all() and n0 == len(mq0) >= n
    assert list(mt0) == list(mt)                                    # same slots, the check only invalidates


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ratio,check,dd", [(1, 0.8, True, 50), (2, 0.7, True, 60), (3, 0.9, False, 50),
                                                 (4, 0.8, True, 256)])
def test_hip_stereo_line_search_matches_oracle(oracle, seed, ratio, check, dd):
    from plvs_amd.linematcher import LineMatcher
    case = make_stereo_case(seed, n_left=100 + 9 * seed, n_right=90 + 5 * seed)
    want_n, wq, wt, wd, wv = oracle_stereo(oracle, case, ratio, check, dd)
    got_n, m, v = LineMatcher(ratio, check).SearchStereoMatchesByKnn(*case, descriptorDist=dd)
    assert got_n == want_n > 5
    assert np.array_equal(m["queryIdx"], wq) and np.array_equal(m["trainIdx"], wt)
    assert np.array_equal(m["distance"], wd) and np.array_equal(v, wv)
    left, ang_l, oct_l, right, ang_r, oct_r = case
    n0, m0, v0 = LineMatcher(ratio, check).SearchStereoMatchesByKnn(left[:0], ang_l[:0], oct_l[:0], right, ang_r, oct_r)
    assert n0 == 0 and len(m0) == 0
