"""The voxblox counterpart of tests/chisel_golden_scenario.py: one fixed sequence run on the reference's own compiled
integrators / mesher (scripts/make_voxblox_golden.py -> tests/golden/voxblox_reference_digests.json), on the oracle and
on the HIP path (tests/test_tsdf_golden_reference.py)."""
import hashlib

import numpy as np

from tests.chisel_golden_scenario import map_digest
from tests.plvs_amd_synth import make_keyframes
from tests.test_tsdf_loadmap import surface_cloud

CASES = [dict(name="simple 5 cm + world cloud", method="simple", vs=0.05, carving=False, far=False, world=True),
         dict(name="merged 5 cm", method="merged", vs=0.05, carving=False, far=False, world=False),
         dict(name="simple 10 cm, carving, depths to 8 m", method="simple", vs=0.10, carving=True, far=True, world=False),
         dict(name="merged 10 cm, carving, depths to 8 m", method="merged", vs=0.10, carving=True, far=True, world=False),
         dict(name="fast 5 cm", method="fast", vs=0.05, carving=False, far=False, world=False),
         dict(name="fast 10 cm, carving, depths to 8 m", method="fast", vs=0.10, carving=True, far=True, world=False)]


def case_inputs(case):
    kfs = make_keyframes(3, max_depth=8.0, room_size=(16.0, 12.0, 3.0), seed=211) if case["far"] else make_keyframes(3, seed=211)
    clouds = []
    for k in kfs:
        xyz = np.ascontiguousarray(k["xyz"][::2], np.float32)
        rgba = np.ascontiguousarray(np.concatenate([k["rgb"][::2], np.full((len(xyz), 1), 255, np.uint8)], 1))
        clouds.append((xyz, rgba, np.ascontiguousarray(k["Twc"], np.float32).reshape(3, 4)))
    world = None
    if case["world"]:
        xyz, rgb, _, nrm = surface_cloud(8000, 213, case["vs"])
        rgba = np.ascontiguousarray(np.concatenate([rgb, np.full((len(rgb), 1), 200, np.uint8)], 1))
        T = np.array([[0.0, -1.0, 0.0, 0.3], [1.0, 0.0, 0.0, -0.2], [0.0, 0.0, 1.0, 0.1]], np.float32)
        world = (xyz, rgba, nrm, T)
    return clouds, world


def inputs_digest():
    h = hashlib.sha1()
    for case in CASES:
        clouds, world = case_inputs(case)
        for c in clouds:
            for a in c:
                h.update(a.tobytes())
        if world:
            for a in world:
                h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def mesh_digest(ids, mesh_block):
    h = hashlib.sha1()
    total = 0
    for bid in sorted(tuple(int(v) for v in b) for b in ids):
        v, n, c = mesh_block(*bid)
        total += len(v)
        for a in (v, n, c):
            h.update(np.ascontiguousarray(a).tobytes())
    return dict(vertices=total, mesh=h.hexdigest())


def run(make_adapter):
    """make_adapter(case) -> object with integrate(xyz, rgba, Twc), world(xyz, rgba, nrm, Twc), block_ids(), get_block(),
    mesh_block(bx, by, bz)."""
    out = []
    for case in CASES:
        a = make_adapter(case)
        clouds, world = case_inputs(case)
        for c in clouds:
            a.integrate(*c)
        if world:
            a.world(*world)
        rec = dict(case=case["name"], **map_digest(a.block_ids(), a.get_block))
        rec.update(mesh_digest(a.block_ids(), a.mesh_block))
        out.append(rec)
    return out
