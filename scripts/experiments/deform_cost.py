"""What Chisel::Deform support costs and how long a deform takes, at the bench's key-frame size (640x480 / 4x4 cloud =
76 800 points... see make_keyframes) and a map of N key frames.  Run on the GPU box:
  python scripts/experiments/deform_cost.py"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from tests.synth_scene import make_keyframes
from plvs_amd.tsdf import TsdfChisel

N = 40
kfs = make_keyframes(N, max_depth=5.0, seed=0)


def run(track, order_free):
    t = TsdfChisel(0.05, max_chunks=65536, order_free=order_free)
    if track:
        t.enable_deform()
    t.integrate(kfs[0]["xyz"], kfs[0]["rgb"], kfs[0]["kfid"], kfs[0]["Twc"])
    t0 = time.perf_counter()
    for k in kfs[1:]:
        t.integrate(k["xyz"], k["rgb"], k["kfid"], k["Twc"])
    dt = (time.perf_counter() - t0) / (N - 1)
    return t, dt


for order_free in (False, True):
    _, plain = run(False, order_free)
    t, tracked = run(True, order_free)
    print("order_free=%d: %.3f ms per key frame (host flavour, upload included), %.3f ms with the chunk order tracked (+%.0f%%)"
          % (order_free, plain * 1e3, tracked * 1e3, 100 * (tracked / plain - 1)))
    kfids = np.arange(N, dtype=np.uint32)
    rng = np.random.default_rng(1)
    Rt = np.zeros((N, 12), np.float32)
    for i in range(N):
        w = rng.normal(scale=0.02, size=3)
        th = np.linalg.norm(w); k = w / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rt[i, :9] = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)).astype(np.float32).reshape(9)
        Rt[i, 9:] = rng.normal(scale=0.05, size=3)
    nch = t.num_chunks()
    t0 = time.perf_counter()
    st = t.deform(kfids, Rt)
    d1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    st2 = t.deform(kfids, Rt)
    d2 = time.perf_counter() - t0
    print("  deform of %d chunks: %.1f ms first call (allocations), %.1f ms second; %d voxels moved -> %d chunks; %.0f M voxels/s"
          % (nch, d1 * 1e3, d2 * 1e3, st["moved"], st["new_chunks"], st2["moved"] / d2 / 1e6))
    t.close()
