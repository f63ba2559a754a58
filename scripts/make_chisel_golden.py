#!/usr/bin/env python3
"""Golden digests of chisel maps built by the REFERENCE ITSELF: the open_chisel library compiled unmodified from
/root/reference (oracle/ref/Makefile -> oracle/_ref/libchisel_full_ref.so) runs tests/chisel_golden_scenario.py and the
per-stage digests go to tests/golden/chisel_reference_digests.json.  Dev-time tool (needs the compiled reference);
tests/test_tsdf_golden_reference.py checks the oracle (CPU) and the HIP path (GPU) against the committed file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import chisel_golden_scenario as S                      # noqa: E402
from tests.test_oracle_pinned_chisel_map import RefChisel          # noqa: E402


class RefAdapter:
    def __init__(self, cam, carving):
        self.m = RefChisel(S.RES, cam, carving=carving)

    def integrate(self, kf, depth):
        self.m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"], depth=depth)

    def world(self, xyz, rgb, kfid, nrm):
        self.m.integrate_world_normals(xyz, rgb, kfid, nrm)

    def deform(self, kfids, Rt):
        self.m.deform(kfids, Rt)

    def digest(self):
        return S.map_digest(self.m.chunk_ids(), self.m.get_chunk)

    def order(self):
        return self.m.chunk_ids()                                  # the container's iteration order

    def meshes(self):
        self.m.update_meshes()
        return S.mesh_digest(self.m.chunk_ids(), self.m.mesh_chunk)


def main():
    inp = S.inputs()
    out = dict(what="sha1 digests of chisel maps built by the reference's own open_chisel sources (see this script)",
               inputs=S.inputs_digest(inp), resolution=S.RES,
               plain=S.run(RefAdapter(inp["cam"], False), inp, False),
               carving=S.run(RefAdapter(inp["cam"], True), inp, True))
    path = os.path.join(ROOT, "tests", "golden", "chisel_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, [r["chunks"] for r in out["plain"]], [r["chunks"] for r in out["carving"]], out["plain"][0]["vertices"])


if __name__ == "__main__":
    main()
