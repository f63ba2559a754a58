"""Lane utilisation of the order-free walk's voxel loop from the rays' step counts: 64 consecutive rays per wave as they come, and
sorted by step count inside their 512-ray tile (VERDICT r3's suggestion).  CPU only."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests.synth_scene import make_keyframes, make_stream_keyframes
# chisel truncation as PLVS sets it (quadratic 0.0019, 0.00152, 0.001504, scale 8): read from the params default
import ctypes
from plvs_amd import _lib
def steps(kf, res=0.05):
    xyz = kf["xyz"]; T = np.asarray(kf["Twc"], np.float64).reshape(3, 4)
    w = xyz @ T[:, :3].T + T[:, 3]
    v = w - T[:, 3]; d = v / np.linalg.norm(v, axis=1, keepdims=True)
    z = xyz[:, 2].astype(np.float64)
    tr = np.maximum((0.0019 * z * z + 0.00152 * z + 0.001504) * 8.0, np.sqrt(3.0) * res)
    s = np.floor((w - d * tr[:, None]) / res); e = np.floor((w + d * tr[:, None]) / res)
    return np.abs(e - s).sum(1).astype(np.int64) + 1
def util(st):
    n = len(st) // 512 * 512
    t = st[:n].reshape(-1, 512)
    a = t.reshape(-1, 8, 64)
    u0 = a.sum() / (a.max(2).sum() * 64)
    ts = np.sort(t, axis=1).reshape(-1, 8, 64)
    u1 = ts.sum() / (ts.max(2).sum() * 64)
    # the tile lives as long as its slowest wave: wave-cycles occupied = 8 * max over the tile
    w0 = a.sum() / (a.max(2).max(1).sum() * 512); w1 = ts.sum() / (ts.max(2).max(1).sum() * 512)
    return round(u0, 3), round(u1, 3), round(w0, 3), round(w1, 3)
for name, kfs in (("room", make_keyframes(6, max_depth=5.0, seed=0)), ("stream", make_stream_keyframes(6, first=300))):
    st = np.concatenate([steps(k) for k in kfs])
    print(name, "mean steps", st.mean().round(2), "lane utilisation as is / sorted in the tile:", util(st)[:2], " vs tile lifetime:", util(st)[2:])
