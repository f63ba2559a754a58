#!/usr/bin/env python3
"""Golden digests of voxblox maps and meshes built by the REFERENCE ITSELF: tsdf_integrator.cc / mesh_integrator.h
compiled unmodified from /root/reference (oracle/ref/Makefile -> oracle/_ref/libvoxblox_ref.so) run
tests/voxblox_golden_scenario.py; the digests go to tests/golden/voxblox_reference_digests.json.  Dev-time tool.  (The
reference takes its pose as a kindr quaternion: the conversion from the pose matrix comes from the oracle's restatement
of that constructor, as in tests/test_oracle_pinned.py.)"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib                                   # noqa: E402
from tests import voxblox_golden_scenario as S                 # noqa: E402

VREF = os.path.join(ROOT, "oracle", "_ref", "libvoxblox_ref.so")
_vp, _i = ctypes.c_void_p, ctypes.c_int


class RefAdapter:
    def __init__(self, case):
        self.ref = ref = ctypes.CDLL(VREF)
        self.ora = oracle_lib.load()
        ref.ref_voxblox_create.restype = _vp
        ref.ref_voxblox_create.argtypes = [ctypes.c_float] * 5 + [_i, ctypes.c_char_p]
        ref.ref_voxblox_integrate.argtypes = [_vp] * 5 + [_i]
        ref.ref_voxblox_integrate_world.argtypes = [_vp] * 6 + [_i]
        ref.ref_voxblox_num_blocks.argtypes = [_vp]
        ref.ref_voxblox_block_ids.argtypes = [_vp, _vp]
        ref.ref_voxblox_get_block.argtypes = [_vp] + [_i] * 3 + [_vp] * 3
        ref.ref_voxblox_mesh_block.argtypes = [_vp] + [_i] * 3 + [_vp] * 3 + [_i]
        self.ora.lib.oracle_voxblox_pose_quat.argtypes = [_vp, _vp]
        self.h = _vp(ref.ref_voxblox_create(case["vs"], 0.1, 10000.0, 0.1, 5.0, int(case["carving"]), case["method"].encode()))

    def pose(self, Twc):
        q = np.zeros(4, np.float32)
        self.ora.lib.oracle_voxblox_pose_quat(Twc.ctypes.data, q.ctypes.data)
        return q, np.ascontiguousarray(Twc[:, 3])

    def integrate(self, xyz, rgba, Twc):
        q, t = self.pose(Twc)
        self.ref.ref_voxblox_integrate(self.h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, len(xyz))

    def world(self, xyz, rgba, nrm, Twc):
        q, t = self.pose(Twc)
        self.ref.ref_voxblox_integrate_world(self.h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, nrm.ctypes.data, len(xyz))
        # the blocks it creates join the layer with the next integratePointCloud call: an empty one
        self.ref.ref_voxblox_integrate(self.h, q.ctypes.data, t.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, 0)

    def block_ids(self):
        n = self.ref.ref_voxblox_num_blocks(self.h)
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.ref.ref_voxblox_block_ids(self.h, ids.ctypes.data)
        return ids[:n]

    def get_block(self, bx, by, bz):
        d, w, c = np.zeros(4096, np.float32), np.zeros(4096, np.float32), np.zeros(4096, np.uint32)
        assert self.ref.ref_voxblox_get_block(self.h, bx, by, bz, d.ctypes.data, w.ctypes.data, c.ctypes.data)
        return d, w, c

    def mesh_block(self, bx, by, bz):
        cap = 4096 * 15
        v, n = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
        c = np.zeros((cap, 4), np.uint8)
        nv = self.ref.ref_voxblox_mesh_block(self.h, bx, by, bz, v.ctypes.data, n.ctypes.data, c.ctypes.data, cap)
        return v[:nv].copy(), n[:nv].copy(), c[:nv].copy()


def main():
    out = dict(what="sha1 digests of voxblox layers and meshes built by the reference's own sources (see this script)",
               inputs=S.inputs_digest(), cases=S.run(RefAdapter))
    path = os.path.join(ROOT, "tests", "golden", "voxblox_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, [(r["chunks"], r["vertices"]) for r in out["cases"]])


if __name__ == "__main__":
    main()
