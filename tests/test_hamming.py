"""Hamming k=2 matching: oracle pinned analytically (CPU), HIP kernel vs oracle (GPU).

Bar: bit-exact train indices and distances, including the tie order.
"""
import numpy as np
import pytest


def _rand_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def _clustered_desc(rng, n, nbase=8, flips=6):
    """Descriptors that are few-bit perturbations of a few bases: many equal
    distances, so the tie rule decides the answer."""
    base = _rand_desc(rng, nbase)
    out = base[rng.integers(0, nbase, size=n)].copy()
    for i in range(n):
        for _ in range(int(rng.integers(0, flips + 1))):
            b = int(rng.integers(0, 256))
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def _np_dist_matrix(q, t):
    x = q[:, None, :] ^ t[None, :, :]
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.int32)


def _np_knn2_lowest(q, t):
    d = _np_dist_matrix(q, t)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    return order.astype(np.int32), np.take_along_axis(d, order, axis=1)


def _np_knn2_mih(q, t):
    """Independent statement of the MIH tie rule: sort by (distance, min per-byte
    popcount s, first byte k reaching s, xor pattern of that byte, train index)."""
    nq, nt = q.shape[0], t.shape[0]
    x = q[:, None, :] ^ t[None, :, :]
    pc = np.unpackbits(x[..., None], axis=3).sum(axis=3)          # nq, nt, 32
    d = pc.sum(axis=2).astype(np.int64)
    s = pc.min(axis=2).astype(np.int64)
    k = pc.argmin(axis=2).astype(np.int64)                         # first minimum
    pat = np.take_along_axis(x, k[..., None], axis=2)[..., 0].astype(np.int64)
    key = (((d * 16 + s) * 32 + k) * 256 + pat) * (1 << 31) + np.arange(nt)[None, :]
    order = np.argsort(key, axis=1, kind="stable")[:, :2]
    return order.astype(np.int32), np.take_along_axis(d, order, axis=1).astype(np.int32)


# ------------------------------------------------------------------ CPU: oracle
def test_oracle_distance_is_popcount(oracle):
    rng = np.random.default_rng(0)
    a, b = _rand_desc(rng, 64), _rand_desc(rng, 64)
    ref = _np_dist_matrix(a, b)
    for i in range(64):
        assert oracle.descriptor_distance(a[i], b[i]) == ref[i, i]
    assert oracle.descriptor_distance(a[0], a[0]) == 0
    assert oracle.descriptor_distance(a[0], ~a[0]) == 256


@pytest.mark.parametrize("nq,nt,seed", [(1, 2, 1), (37, 53, 2), (100, 100, 3), (64, 700, 4)])
def test_oracle_bf_matches_numpy(oracle, nq, nt, seed):
    rng = np.random.default_rng(seed)
    q, t = _clustered_desc(rng, nq), _clustered_desc(rng, nt)
    idx, dist = oracle.knn2(q, t, mih=False)
    ridx, rdist = _np_knn2_lowest(q, t)
    assert np.array_equal(dist, rdist)
    assert np.array_equal(idx, ridx)


@pytest.mark.parametrize("nq,nt,seed,clustered", [(1, 2, 1, True), (37, 53, 2, True),
                                                   (100, 100, 3, True), (50, 300, 4, False),
                                                   (64, 900, 5, True)])
def test_oracle_mih_matches_stated_tie_rule(oracle, nq, nt, seed, clustered):
    """The restated Mihasher (hash tables, radius growth, early exit) must give
    exact distances and the closed-form discovery order."""
    rng = np.random.default_rng(seed)
    gen = _clustered_desc if clustered else _rand_desc
    q, t = gen(rng, nq), gen(rng, nt)
    idx, dist = oracle.knn2(q, t, mih=True)
    ridx, rdist = _np_knn2_mih(q, t)
    assert np.array_equal(dist, rdist)
    assert np.array_equal(idx, ridx)


def test_oracle_mask_and_degenerate(oracle):
    rng = np.random.default_rng(7)
    q, t = _rand_desc(rng, 5), _rand_desc(rng, 1)
    mask = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
    for mih in (False, True):
        idx, dist = oracle.knn2(q, t, mask, mih=mih)
        assert np.all(idx[mask == 0] == -1) and np.all(dist[mask == 0] == -1)
        assert np.all(idx[mask == 1, 0] == 0) and np.all(idx[mask == 1, 1] == -1)


# ------------------------------------------------------------- GPU: HIP kernel
@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,seed,clustered", [(1, 1, 0, False), (1, 2, 1, True), (3, 64, 2, True),
                                                   (100, 100, 3, True), (257, 1000, 4, True),
                                                   (2000, 2000, 5, False), (500, 2000, 6, True)])
@pytest.mark.parametrize("mih", [False, True])
def test_hip_knn2_matches_oracle(oracle, nq, nt, seed, clustered, mih):
    from plvs_amd import _lib
    from plvs_amd.matcher import knn2_raw
    rng = np.random.default_rng(seed)
    gen = _clustered_desc if clustered else _rand_desc
    q, t = gen(rng, nq), gen(rng, nt)
    mask = None if seed % 2 == 0 else (rng.integers(0, 4, size=nq) > 0).astype(np.uint8)
    rule = _lib.TIE_MIH if mih else _lib.TIE_LOWEST_INDEX
    idx, dist = knn2_raw(q, t, mask, rule)                         # host flavour
    oidx, odist = oracle.knn2(q, t, mask, mih=mih)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)

    import torch                                                    # device flavour
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    dm = None if mask is None else torch.from_numpy(mask).cuda()
    didx, ddist = knn2_raw(dq, dt, dm, rule)
    torch.cuda.synchronize()
    assert np.array_equal(didx.cpu().numpy(), oidx)
    assert np.array_equal(ddist.cpu().numpy(), odist)


@pytest.mark.gpu
def test_hip_knn2_errors_like_reference(capsys):
    from plvs_amd import BinaryDescriptorMatcher, _lib
    from plvs_amd.matcher import knn2_raw
    m = BinaryDescriptorMatcher()
    out = []
    m.knnMatch(np.zeros((0, 32), np.uint8), np.zeros((4, 32), np.uint8), out)
    assert out == [] and "cannot be void" in capsys.readouterr().out
    m.knnMatch(np.zeros((3, 32), np.uint8), np.zeros((4, 32), np.uint8), out, mask=np.ones((2, 1), np.uint8))
    assert out == [] and "input mask should have 3 rows" in capsys.readouterr().out
    with pytest.raises(_lib.PlvsHipError) as e:
        knn2_raw(np.zeros((0, 32), np.uint8), np.zeros((4, 32), np.uint8))
    assert e.value.code == _lib.PLVS_ERR_EMPTY


@pytest.mark.gpu
def test_hip_knnmatch_mirror_compact(oracle):
    from plvs_amd import BinaryDescriptorMatcher
    rng = np.random.default_rng(11)
    q, t = _clustered_desc(rng, 40), _clustered_desc(rng, 60)
    mask = (rng.integers(0, 3, size=(40, 1)) > 0).astype(np.uint8)
    out = []
    BinaryDescriptorMatcher().knnMatch(q, t, out, 2, mask, True)
    oidx, odist = oracle.knn2(q, t, mask, mih=True)
    kept = [i for i in range(40) if mask[i, 0]]
    assert len(out) == len(kept)
    for row, i in zip(out, kept):
        assert [m.trainIdx for m in row] == list(oidx[i])
        assert [m.distance for m in row] == [float(x) for x in odist[i]]
        assert all(m.queryIdx == i and m.imgIdx == 0 for m in row)
