"""One fixed sequence of chisel map operations, run on three implementations through thin adapters: the reference's own
compiled library (scripts/make_chisel_golden.py -> tests/golden/chisel_reference_digests.json), the oracle and the HIP
path (tests/test_tsdf_golden_reference.py).  After every stage the whole map is digested: sorted chunk ids and the
sdf / weight / kfid / colour planes of every chunk, plus the chunk container's iteration order."""
import hashlib

import numpy as np

from tests.plvs_amd_synth import TUM1, make_keyframes
from tests.test_tsdf_loadmap import surface_cloud

RES = 0.05


def cam():
    c = dict(TUM1)
    for k in ("fx", "fy", "cx", "cy"):
        c[k] = c[k] / 4
    c["width"] //= 4
    c["height"] //= 4
    return c


def motions(kfids, seed, rot, shift):
    rng = np.random.default_rng(seed)
    out = np.zeros((len(kfids), 12), np.float32)
    for i in range(len(kfids)):
        w = rng.normal(scale=rot, size=3)
        th = np.linalg.norm(w)
        k = w / max(th, 1e-12)
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        out[i, :9] = R.astype(np.float32).reshape(9)
        out[i, 9:] = rng.normal(scale=shift, size=3).astype(np.float32)
    return out


def depth_image(c, seed):
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.4, 4.8, (c["height"], c["width"])).astype(np.float32)
    d[rng.random(d.shape) < 0.15] = np.nan
    return d


def inputs():
    c = cam()
    kfs = make_keyframes(5, cam=c, seed=101)
    kfids = np.unique(np.concatenate([k["kfid"] for k in kfs]))
    world = surface_cloud(2500, seed=103)
    return dict(cam=c, kfs=kfs, kfids=kfids, world=(world[0], world[1], world[2] % np.uint32(5), world[3]),
                Rt1=motions(kfids, 105, 0.03, 0.08), Rt2=motions(kfids[1:], 107, 0.2, 0.5),
                depths=[depth_image(c, 109 + i) for i in range(4)])


def inputs_digest(inp):
    h = hashlib.sha1()
    for k in inp["kfs"]:
        for name in ("xyz", "rgb", "kfid", "Twc"):
            h.update(np.ascontiguousarray(k[name]).tobytes())
    for a in inp["world"] + (inp["Rt1"], inp["Rt2"]) + tuple(inp["depths"]):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def map_digest(chunk_ids, get_chunk):
    h = hashlib.sha1()
    ids = sorted(tuple(int(v) for v in c) for c in chunk_ids)
    for cid in ids:
        h.update(np.array(cid, np.int32).tobytes())
        for a in get_chunk(*cid):
            h.update(np.ascontiguousarray(a).tobytes())
    return dict(chunks=len(ids), planes=h.hexdigest())


def order_digest(order):
    return hashlib.sha1(np.ascontiguousarray(order, np.int32).tobytes()).hexdigest()


def mesh_digest(ids, mesh_chunk):
    h = hashlib.sha1()
    total = 0
    for cid in sorted(tuple(int(v) for v in c) for c in ids):
        v, n, c, k = mesh_chunk(*cid)
        total += len(k)
        for a in (v, n, c, k):
            h.update(np.ascontiguousarray(a).tobytes())
    return dict(vertices=total, mesh=h.hexdigest())


def run(a, inp, carving):
    """a: adapter with integrate(kf, depth), world(xyz, rgb, kfid, nrm), deform(kfids, Rt), digest(), order(), meshes().
    -> list of per-stage records."""
    out = []

    def stage(name, with_mesh=False):
        rec = dict(stage=name, **a.digest())
        rec["order"] = order_digest(a.order())
        if with_mesh:
            rec.update(a.meshes())
        out.append(rec)

    kfs = inp["kfs"]
    if carving:                                   # IntegratePointCloudWidthDepth with carving: every call brings a depth image
        for i in range(4):
            a.integrate(kfs[i], inp["depths"][i])
            stage(f"carve+integrate {i}")
        return out
    for i in range(3):
        a.integrate(kfs[i], None)
    stage("3 key frames", with_mesh=True)
    a.world(*inp["world"])
    stage("world cloud with normals")
    a.deform(inp["kfids"], inp["Rt1"])
    stage("deform (small correction)")
    a.integrate(kfs[3], None)
    a.integrate(kfs[4], None)
    stage("2 more key frames")
    a.deform(inp["kfids"][1:], inp["Rt2"])
    stage("deform (large correction, one key frame dropped)")
    return out
