import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from tests.synth_scene import make_keyframes
from plvs_amd.tsdf import TsdfChisel
from tests.test_shard_rays import WIDTHS, send_buffers, virtual_all_to_all
kfs = make_keyframes(100, max_depth=5.0, seed=0)
xyz = torch.from_numpy(np.concatenate([k["xyz"] for k in kfs])).cuda()
rgb = torch.from_numpy(np.concatenate([k["rgb"] for k in kfs])).cuda()
kfid = torch.from_numpy(np.concatenate([k["kfid"] for k in kfs]).astype(np.int32)).cuda()
Twc = torch.from_numpy(np.stack([k["Twc"] for k in kfs])).cuda()
offsets = np.cumsum([0] + [k["xyz"].shape[0] for k in kfs]).astype(np.int32)
t = TsdfChisel(0.05, max_chunks=16384, shard_rank=0, shard_count=1, order_free=True)
for lap in range(10):
    c = t.shard_walk(xyz, offsets, Twc)
    bufs = send_buffers(t, c)
    torch.cuda.synchronize()
    t.shard_apply(bufs[0], bufs[1], bufs[2], c.reshape(1,3), rgb, kfid)
    sat = t.shard_saturated()
    if sat.shape[0]: t.shard_note_saturated(sat)
    print("lap %d: descriptors %d, voxel sums %d, colour-run records %d (%.1f MB), newly saturated %d" % (lap, c[0,0], c[0,1], c[0,2], c[0,2]*24/1e6, sat.shape[0]))
