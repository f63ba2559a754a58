"""Stage times of the order-free integrate on the bench stream for developer builds of the library
(make -C plvs_amd/csrc variant NAME=... DEFS=...): one subprocess per library (PLVS_HIP_LIB).
`prof` prints the phase clocks of walk_tiles (share of the summed tile cycles per phase).
usage: walk_variants.py [name ...]      e.g. walk_variants.py base prof exp1 exp2"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PHASES = ["setup", "walk(thread0)", "walk(wait longest)", "entries->chunk cache", "cache->directory", "ranks+colour weights",
          "records", "runs", "epilogue"]


def child():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from plvs_amd import _lib
    from tests.synth_scene import make_keyframes
    from plvs_amd.tsdf import TsdfChisel
    kfs = make_keyframes(100, max_depth=5.0, seed=0)
    xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
    rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
    kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
    Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
    offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
    t = TsdfChisel(0.05, max_chunks=16384, order_free=True)
    for _ in range(6):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    prof = hasattr(_lib.lib, "plvs_hip_debug_walk_prof")
    if prof:
        _lib.lib.plvs_hip_debug_walk_prof(None, 1)
    t.set_profiling(True)
    for _ in range(6):
        t.integrate_batch_dev(xyz, rgb, kfid, offsets, Twc)
    sm, c = t.stage_ms()
    out = {"stage_ms": {k: round(v / c, 4) for k, v in sm.items()}, "visits": t.last_stats()["visits"]}
    if prof:
        buf = (ctypes.c_ulonglong * 16)()
        _lib.lib.plvs_hip_debug_walk_prof(buf, 0)
        tot = float(sum(buf[:9])) or 1.0
        out["phase_share"] = {PHASES[i]: round(buf[i] / tot, 4) for i in range(9)}
    t.close()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("WALK_VARIANT_CHILD"):
        child()
        sys.exit(0)
    for name in sys.argv[1:] or ["base"]:
        env = dict(os.environ, WALK_VARIANT_CHILD="1")
        if name != "base":
            env["PLVS_HIP_LIB"] = os.path.join(ROOT, "plvs_amd", "lib", f"libplvs_hip_{name}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(name, lines[-1][7:] if lines else ("FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])), flush=True)
