import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from tests.synth_scene import make_keyframes
from plvs_amd.tsdf import TsdfVoxblox
kfs = make_keyframes(20, max_depth=5.0, seed=0)
for meth in ("simple", "merged"):
    t = TsdfVoxblox(0.05, max_blocks=65536)
    rg = [np.concatenate([k["rgb"], np.full((len(k["rgb"]),1),255,np.uint8)],1) for k in kfs]
    f = t.integrate if meth=="simple" else t.integrate_merged
    f(kfs[0]["xyz"], rg[0], kfs[0]["Twc"])
    t0=time.perf_counter(); v=0
    for k,c in zip(kfs[1:], rg[1:]):
        f(k["xyz"], c, k["Twc"]); v+=t.last_stats()["visits"]
    dt=(time.perf_counter()-t0)/19
    print(meth, "%.2f ms per 76 800-point keyframe (host flavour, upload included), %.2f M voxel updates per keyframe" % (dt*1e3, v/19/1e6))
