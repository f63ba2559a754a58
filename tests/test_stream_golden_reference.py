"""Steps 14 .. 25 of the streaming headline (bench.py's default workload: 2 500 key frames of the office loop, chisel 5 cm /
5 m, depth images straight into an order-free map) against golden data the REFERENCE ITSELF made (scripts/make_stream_golden.py:
open_chisel compiled unmodified, the 2 500 clouds one InsertCloud at a time):

  every step 14 .. 25   the exact part of the map — chunk set, observed voxels, key-frame ids, colours — by sha256;
  steps 19 and 25       sdf / weight of 4 096 sampled voxels within the order-free bound of the reference's f32 values
                        (a voxel with n visits: 2e-5 m + n 2^-24 tau, 5e-5 + n 2^-24: tests/test_measured_configs.py).

Steps 1 .. 5 against the oracle voxel by voxel and 1 .. 13 against the point-stream entry point: tests/test_measured_configs.py."""
import json
import os

import numpy as np
import pytest

from tests import stream_golden_scenario as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    with open(os.path.join(GOLDEN, "stream_reference_steps.json")) as f:
        steps = json.load(f)["steps"]
    return steps, np.load(os.path.join(GOLDEN, "stream_reference_samples.npz"))


def test_golden_files_cover_the_steps():
    steps, samples = _golden()
    assert sorted(int(k) for k in steps) == list(range(S.CHECK_FROM, S.STEPS + 1))
    for st in S.SAMPLE_STEPS:
        assert samples[f"vox_{st}"].shape == (S.NSAMPLE,) and (samples[f"w_{st}"] > 0).all()


@pytest.mark.gpu
def test_hip_stream_steps_14_to_25_reproduce_the_reference_made_digests():
    from plvs_amd.tsdf import TsdfChisel
    from tests.test_measured_configs import _depth_batch
    steps, samples = _golden()
    tau = max((0.0019 * 25.0 - 0.00152 * 5.0 + 0.001504) * 6.0, 2.0 * np.sqrt(3.0) * S.RES)
    w_min = 1.0 / (2.0 * tau)
    dev = TsdfChisel(S.RES, max_chunks=16384, order_free=True)     # as bench.py creates it
    for step in range(1, S.STEPS + 1):
        dev.integrate_depth_batch_dev(*_depth_batch(S.keyframes(step, images=True), max_depth=S.MAX_DEPTH))
        if step < S.CHECK_FROM:
            continue
        chunks = {}

        def get(cx, cy, cz):
            if (cx, cy, cz) not in chunks:
                chunks[(cx, cy, cz)] = dev.get_chunk(cx, cy, cz)
            return chunks[(cx, cy, cz)]
        digest, n = S.exact_digest(dev.chunk_ids(), get)
        assert n == steps[str(step)]["chunks"] and digest == steps[str(step)]["exact_sha256"], f"step {step}"
        if step in S.SAMPLE_STEPS:
            ids, vox = samples[f"ids_{step}"], samples[f"vox_{step}"]
            want_s, want_w = samples[f"sdf_{step}"], samples[f"w_{step}"]
            got_s = np.array([get(*map(int, c))[0][v] for c, v in zip(ids, vox)], np.float32)
            got_w = np.array([get(*map(int, c))[1][v] for c, v in zip(ids, vox)], np.float32)
            n_up = np.ceil(want_w.astype(np.float64) / w_min) + 2.0
            worst_s = float((np.abs(got_s.astype(np.float64) - want_s) / (2e-5 + n_up * 2.0 ** -24 * tau)).max())
            worst_w = float((np.abs(got_w.astype(np.float64) - want_w) / want_w / (5e-5 + n_up * 2.0 ** -24)).max())
            print(f"step {step}: worst fraction of the bound: sdf {worst_s:.3f}, weight {worst_w:.3f}")
            assert worst_s <= 1.0 and worst_w <= 1.0, (step, worst_s, worst_w)
    dev.close()
