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
