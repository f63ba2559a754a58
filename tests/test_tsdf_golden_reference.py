"""Golden vectors from the reference itself: tests/golden/chisel_reference_digests.json holds, stage by stage, digests of
the maps the reference's own open_chisel sources built (scripts/make_chisel_golden.py: the compiled reference library
in this container; it does not exist on the GPU box).  The oracle (CPU) and the HIP path (GPU, through the C ABI, the
bit-exact mode) run the same sequence — key frames, a world cloud along its normals, carving by depth images, two
Chisel::Deform calls, chunk meshes — and must reproduce every digest: all voxel planes of all chunks, the chunk
container's iteration order, every mesh vertex."""
import json
import os

import numpy as np
import pytest

from tests import chisel_golden_scenario as S
from tests import oracle_lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chisel_reference_digests.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        g = json.load(f)
    inp = S.inputs()
    assert S.inputs_digest(inp) == g["inputs"], "the synthetic inputs changed: regenerate with scripts/make_chisel_golden.py"
    return g, inp


class OracleAdapter:
    def __init__(self, cam):
        self.cam = cam
        self.m = oracle_lib.load().chisel(S.RES).track_order()

    def integrate(self, kf, depth):
        c = self.cam
        if depth is not None:
            self.m.carve(depth, c["fx"], c["fy"], c["cx"], c["cy"], kf["Twc"])
        self.m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
        self.m.end_call()

    def world(self, xyz, rgb, kfid, nrm):
        self.m.integrate_world_normals(xyz, rgb, kfid, nrm)
        self.m.end_call()

    def deform(self, kfids, Rt):
        assert self.m.deform(kfids, Rt)[2] == 0

    def digest(self):
        return S.map_digest(self.m.chunk_ids(), self.m.get_chunk)

    def order(self):
        return self.m.chunk_order()

    def meshes(self):
        return S.mesh_digest(self.m.chunk_ids(), self.m.mesh_chunk)


class DeviceAdapter:
    def __init__(self, cam):
        from plvs_amd.tsdf import TsdfChisel
        self.cam = cam
        self.m = TsdfChisel(S.RES, max_chunks=4096).enable_deform()

    def integrate(self, kf, depth):
        c = self.cam
        if depth is not None:
            self.m.carve(depth, c["fx"], c["fy"], c["cx"], c["cy"], kf["Twc"])
        self.m.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])

    def world(self, xyz, rgb, kfid, nrm):
        self.m.integrate_world_normals(xyz, rgb, kfid, nrm)

    def deform(self, kfids, Rt):
        assert self.m.deform(kfids, Rt)["undefined"] == 0

    def digest(self):
        return S.map_digest(self.m.chunk_ids(), self.m.get_chunk)

    def order(self):
        return self.m.chunk_order()

    def meshes(self):
        ids = sorted(tuple(int(v) for v in c) for c in self.m.chunk_ids())
        r = self.m.mesh_chunks(np.array(ids, np.int32))
        first = r["chunk_first"]
        per = {cid: tuple(r[k][int(first[i]):int(first[i + 1])] for k in ("vertices", "normals", "colors", "kfids"))
               for i, cid in enumerate(ids)}
        return S.mesh_digest(ids, lambda *cid: per[tuple(cid)])


def check(adapter, golden, carving):
    g, inp = golden
    got = S.run(adapter, inp, carving)
    want = g["carving" if carving else "plain"]
    assert [r["stage"] for r in got] == [r["stage"] for r in want]
    for a, b in zip(got, want):
        assert a == b, f"stage '{a['stage']}' differs from the reference's map: {a} vs {b}"
    return got


@pytest.mark.parametrize("carving", [False, True])
def test_oracle_reproduces_the_reference_built_maps(golden, carving):
    got = check(OracleAdapter(golden[1]["cam"]), golden, carving)
    assert got[-1]["chunks"] > 8


@pytest.mark.gpu
@pytest.mark.parametrize("carving", [False, True])
def test_hip_reproduces_the_reference_built_maps(golden, carving):
    a = DeviceAdapter(golden[1]["cam"])
    got = check(a, golden, carving)
    assert got[-1]["chunks"] > 8
    a.m.close()


# ------------------------------------------------------------------ voxblox
from tests import voxblox_golden_scenario as V   # noqa: E402

VGOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxblox_reference_digests.json")


@pytest.fixture(scope="module")
def vgolden():
    with open(VGOLDEN) as f:
        g = json.load(f)
    assert V.inputs_digest() == g["inputs"], "the synthetic inputs changed: regenerate with scripts/make_voxblox_golden.py"
    return g


class VoxbloxOracleAdapter:
    def __init__(self, case):
        self.m = oracle_lib.load().voxblox(case["vs"], carving=case["carving"])
        self.method = case["method"]

    def integrate(self, xyz, rgba, Twc):
        if self.method == "fast":
            self._fast(xyz, rgba, Twc)
        else:
            (self.m.integrate_merged if self.method == "merged" else self.m.integrate)(xyz, rgba, Twc)

    def _fast(self, xyz, rgba, Twc):   # the reference's approximate sets (the oracle also has collision-free ones)
        self.m.integrate_fast(xyz, rgba, Twc, approx_sets=True)

    def world(self, xyz, rgba, nrm, Twc):
        self.m.integrate_world_normals(xyz, rgba, nrm, Twc)

    def block_ids(self):
        return self.m.chunk_ids()

    def get_block(self, bx, by, bz):
        return self.m.get_chunk(bx, by, bz)

    def mesh_block(self, bx, by, bz):
        from tests.test_tsdf_voxblox_mesh import mesh_block
        return mesh_block(self.m, bx, by, bz)


class VoxbloxDeviceAdapter(VoxbloxOracleAdapter):
    def __init__(self, case):
        from plvs_amd.tsdf import TsdfVoxblox
        self.m = TsdfVoxblox(case["vs"], use_carving=case["carving"], max_blocks=8192)
        self.method = case["method"]
        self._mesh = None

    def _fast(self, xyz, rgba, Twc):
        self.m.integrate_fast(xyz, rgba, Twc)

    def mesh_block(self, bx, by, bz):
        if self._mesh is None:
            ids = sorted(tuple(int(v) for v in b) for b in self.m.chunk_ids())
            r = self.m.mesh_blocks(np.array(ids, np.int32))
            first = r["block_first"]
            self._mesh = {b: tuple(r[k][int(first[i]):int(first[i + 1])] for k in ("vertices", "normals", "colors"))
                          for i, b in enumerate(ids)}
        return self._mesh[(bx, by, bz)]


def vcheck(make_adapter, g):
    got = V.run(make_adapter)
    assert [r["case"] for r in got] == [r["case"] for r in g["cases"]]
    for a, b in zip(got, g["cases"]):
        assert a == b, f"case '{a['case']}' differs from the reference's layer / mesh: {a} vs {b}"
        assert a["chunks"] > 20 and a["vertices"] > 5000


def test_oracle_reproduces_the_reference_built_voxblox_layers(vgolden):
    vcheck(VoxbloxOracleAdapter, vgolden)


@pytest.mark.gpu
def test_hip_reproduces_the_reference_built_voxblox_layers(vgolden):
    vcheck(VoxbloxDeviceAdapter, vgolden)
