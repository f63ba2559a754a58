"""Dense stereo by semi-global matching (sgm::StereoSGM as PLVS uses it): the oracle's
properties on CPU, and the HIP path against the oracle stage by stage, bit for bit, through the C ABI."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.oracle_lib import golden


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def kitti_pair(width=1240, height=376, x0=0, y0=0):
    left, right = golden("urban1_1241x376.pgm"), golden("urban1_right_1241x376.pgm")
    return (np.ascontiguousarray(left[y0:y0 + height, x0:x0 + width]),
            np.ascontiguousarray(right[y0:y0 + height, x0:x0 + width]))


def test_oracle_recovers_a_known_shift(oracle):
    """right = left shifted by 23 px: the disparity is 23 wherever it is defined."""
    img = golden("aloe_640x480.pgm")
    left, right = np.ascontiguousarray(img[:240, :400]), np.ascontiguousarray(img[:240, 23:423])
    disp = oracle.sgm(left, right)
    inner = disp[20:-20, 100:-20]
    assert (inner > 0).mean() > 0.9
    assert (np.abs(inner[inner > 0].astype(int) - 23) <= 1).mean() > 0.98
    # the 16-pixel remainder of check_consistency's grid and the median's border
    assert (disp[0] == 0).all() and (disp[:, 0] == 0).all() and (disp[-1] == 0).all()


def test_oracle_agrees_with_the_sparse_stereo_matcher(oracle):
    """Two independent estimators on a real pair: the dense disparity at an ORB keypoint vs uL - uR of
    Frame::ComputeStereoMatches (oracle/stereo.c)."""
    from tests.test_stereo import KITTI_BF, MB, oracle_side, scale_tables
    left, right = kitti_pair()
    disp, st = oracle.sgm(left, right, stages=True)
    assert (disp > 0).mean() > 0.7
    (kl, dl, pl), (kr, dr, pr) = oracle_side(oracle, left, right, 2000)
    s, inv = scale_tables()
    u, z, score, kept = oracle.stereo_matches(kl, dl, kr, dr, pl, pr, s, inv, MB, np.float32(KITTI_BF))
    ok = (u >= 0) & (kl["octave"] <= 2)
    sparse = kl["x"][ok] - u[ok]
    dense = disp[np.round(kl["y"][ok]).astype(int), np.round(kl["x"][ok]).astype(int)].astype(np.float32)
    both = (dense > 0) & (sparse < 60) & (sparse > 2)
    assert both.sum() > 150
    assert (np.abs(dense[both] - sparse[both]) <= 2.0).mean() > 0.85
    # stage invariants: census is 31 bits and zero on the border; the summed costs of eight paths fit 8 * (P2 + 31)
    assert (st["census_left"] < 2 ** 31).all() and (st["census_left"][:3] == 0).all() and (st["census_left"][:, :4] == 0).all()
    assert st["cost_sum"].max() <= 8 * (120 + 31)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["kitti_1240x376", "odd_333x181", "small_64x48", "flat"])
def test_hip_sgm_matches_oracle(oracle, case):
    from plvs_amd.sgm import StereoSGM
    if case == "kitti_1240x376":
        left, right = kitti_pair()
    elif case == "odd_333x181":
        left, right = kitti_pair(333, 181, 500, 100)
    elif case == "small_64x48":
        left, right = kitti_pair(64, 48, 300, 200)
    else:
        left = np.full((64, 96), 128, np.uint8)
        left[20:40, 30:60] = 0                       # zero pixels are masked by the consistency check
        right = left.copy()
    h, w = left.shape
    want, st = oracle.sgm(left, right, stages=True)
    sgm = StereoSGM(w, h)
    got = sgm.execute(left, right)
    for name in ("census_left", "census_right", "cost_sum", "raw_left", "raw_right", "median_left", "median_right"):
        assert np.array_equal(sgm.stage(name), st[name]), name
    assert np.array_equal(got, want)
    # other parameters, and the device flavour
    import torch
    p = StereoSGM.Parameters(P1=7, P2=60, uniqueness=0.9)
    sgm2 = StereoSGM(w, h, param=p)
    d_out = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    sgm2.execute_dev(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), d_out)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), oracle.sgm(left, right, 7, 60, 0.9))
    with pytest.raises(ValueError):
        StereoSGM(w, h, disparity_size=32)
