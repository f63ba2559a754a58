#!/usr/bin/env python3
"""Golden data of the streaming headline made by the REFERENCE ITSELF: open_chisel compiled unmodified from /root/reference
(oracle/ref/Makefile -> oracle/_ref/libchisel_full_ref_o3.so) integrates the 2 500 key-frame clouds of tests/stream_golden_scenario.py
one by one, as PointCloudMapChisel::InsertCloud would; digests and samples of steps 14 .. 25 go to tests/golden/.  Dev-time tool
(needs the compiled reference; ~2 minutes); tests/test_stream_golden_reference.py checks the HIP path against the committed files."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import stream_golden_scenario as S                      # noqa: E402
from tests.synth_scene import TUM1                                 # noqa: E402
from tests.test_oracle_pinned_chisel_map import RefChisel          # noqa: E402


def main():
    ref = RefChisel(S.RES, dict(TUM1))
    steps, samples = {}, {}
    t0 = time.time()
    for step in range(1, S.STEPS + 1):
        visits_before = None
        for kf in S.keyframes(step):
            ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        if step >= S.CHECK_FROM:
            dg, n = S.exact_digest(ref.chunk_ids(), ref.get_chunk)
            steps[str(step)] = {"exact_sha256": dg, "chunks": n}
        if step in S.SAMPLE_STEPS:
            ids, vox = S.sample_positions(ref.chunk_ids(), ref.get_chunk, step)
            sdf = np.zeros(len(vox), np.float32)
            w = np.zeros(len(vox), np.float32)
            cache = {}
            for i, (cid, v) in enumerate(zip(ids, vox)):
                cid = tuple(int(x) for x in cid)
                if cid not in cache:
                    cache[cid] = ref.get_chunk(*cid)
                sdf[i], w[i] = cache[cid][0][v], cache[cid][1][v]
            samples[f"ids_{step}"], samples[f"vox_{step}"], samples[f"sdf_{step}"], samples[f"w_{step}"] = ids, vox, sdf, w
        print(f"step {step}: {ref.num_chunks()} chunks, {time.time() - t0:.0f} s", flush=True)
    out = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(out, "stream_reference_steps.json"), "w") as f:
        json.dump({"made_by": "scripts/make_stream_golden.py (the reference's open_chisel, compiled unmodified)", "steps": steps}, f, indent=1)
    np.savez_compressed(os.path.join(out, "stream_reference_samples.npz"), **samples)


if __name__ == "__main__":
    main()
