#!/usr/bin/env python3
"""What does voxblox's FastTsdfIntegrator — PLVS's YAML default, `voxbloxIntegrationMethod: "fast"`,
src/PointCloudMapVoxblox.cc:44 — put into a map, compared with the Simple and the Merged integrator the HIP path
implements?  All three are the reference's OWN code (tsdf_integrator.cc compiled unmodified, oracle/_ref/libvoxblox_ref.so,
one integrator thread), run over the same clouds.  Dev-time tool; writes profiles/r03_voxblox_fast_vs_simple.json,
quoted by INTEGRATION.md §4.

`fast` is a lossy speed-up of `simple`: a point whose 2x-finer start voxel was already seen in this scan casts no ray
at all, a ray stops after more than two consecutive voxels that some ray of this scan has already updated, and it walks
from the surface back towards the camera.  Every update it does make is updateTsdfVoxel — the same arithmetic — so the
maps agree on WHAT a voxel update is and differ in HOW MANY updates a voxel receives."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.make_voxblox_golden import RefAdapter                       # noqa: E402
from tests.plvs_amd_synth import make_keyframes                          # noqa: E402


def clouds(n, far, stride):
    kfs = make_keyframes(n, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=211) if far else make_keyframes(n, seed=211)
    out = []
    for k in kfs:
        xyz = np.ascontiguousarray(k["xyz"][::stride], np.float32)
        rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::stride], np.full((len(xyz), 1), 255, np.uint8)], 1))
        out.append((xyz, rgba, np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)))
    return out


def build(method, vs, carving, cl):
    a = RefAdapter(dict(vs=vs, carving=carving, method=method))
    t0 = time.perf_counter()
    for c in cl:
        a.integrate(*c)
    dt = time.perf_counter() - t0
    m = {}
    for b in a.block_ids():
        m[tuple(int(v) for v in b)] = a.get_block(*b)
    return m, dt


def compare(ref, other, trunc):
    """ref / other: {block: (distance, weight, rgba)}.  Voxels observed (weight > 0) by both, by one only; on the common
    ones |d distance| and the weight ratio."""
    both = only_ref = only_other = 0
    dd, wr = [], []
    for b in set(ref) | set(other):
        kr = ref[b][1] > 1e-6 if b in ref else np.zeros(4096, bool)
        ko = other[b][1] > 1e-6 if b in other else np.zeros(4096, bool)
        c = kr & ko
        both += int(c.sum()); only_ref += int((kr & ~ko).sum()); only_other += int((ko & ~kr).sum())
        if c.any():
            dd.append(np.abs(ref[b][0][c] - other[b][0][c]))
            wr.append(other[b][1][c] / ref[b][1][c])
    dd = np.concatenate(dd) if dd else np.zeros(1)
    wr = np.concatenate(wr) if wr else np.ones(1)
    return dict(observed_by_both=both, only_in_first=only_ref, only_in_second=only_other,
                abs_ddistance_m=dict(median=float(np.median(dd)), p90=float(np.percentile(dd, 90)),
                                     p99=float(np.percentile(dd, 99)), max=float(dd.max()),
                                     fraction_above_a_tenth_of_truncation=float((dd > 0.1 * trunc).mean())),
                weight_ratio_second_over_first=dict(median=float(np.median(wr)), p10=float(np.percentile(wr, 10)),
                                                    p90=float(np.percentile(wr, 90))))


def main():
    out = dict(what=__doc__.split("\n\n")[0], truncation_m=0.1, cases=[])
    for name, vs, carving, far, n, stride in (("5 cm, 6 key frames, no carving", 0.05, False, False, 6, 1),
                                              ("5 cm, 6 key frames, carving", 0.05, True, False, 6, 1),
                                              ("2 cm, 3 key frames (configs[3] room, depths to 8 m), no carving", 0.02, False, True, 3, 1)):
        cl = clouds(n, far, stride)
        maps, secs = {}, {}
        for method in ("simple", "merged", "fast"):
            maps[method], secs[method] = build(method, vs, carving, cl)
        rec = dict(case=name, points=int(sum(len(c[0]) for c in cl)),
                   blocks={m: len(maps[m]) for m in maps},
                   seconds_one_thread={m: round(secs[m], 3) for m in secs},
                   simple_vs_fast=compare(maps["simple"], maps["fast"], 0.1),
                   simple_vs_merged=compare(maps["simple"], maps["merged"], 0.1),
                   merged_vs_fast=compare(maps["merged"], maps["fast"], 0.1))
        out["cases"].append(rec)
        print(json.dumps(rec, indent=1))
    path = os.path.join(ROOT, "profiles", "r03_voxblox_fast_vs_simple.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path)


if __name__ == "__main__":
    main()
