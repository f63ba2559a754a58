"""How far does a build of the reference with ITS OWN flags (-O3 -march=..., gcc's default -ffp-contract=fast: FMA
contraction) drift from the no-FMA build the oracle is pinned against?  Container with /root/reference only:
  cd oracle/ref && g++ -O3 -march=x86-64-v3 -std=c++14 -fPIC -w -Ieigen_full -I$CH/include -shared \
      chisel_full_ref_wrap.cpp $CH/src/*.cpp $CH/src/*/*.cpp -o /tmp/libchisel_full_fast.so -lpthread
  python scripts/experiments/reference_fma_build_probe.py
Result (5 key frames, 5 cm): same 24 chunks; 14 257 of 17 452 known voxels differ in the sdf's last bits, by at most
8.1e-7 m; weights, kfids, colours identical."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tests.test_oracle_pinned_chisel_map as T
from tests import oracle_lib
from tests.plvs_amd_synth import make_keyframes
T.REF = '/tmp/libchisel_full_fast.so'
ora_lib = oracle_lib.load()
cam = T.small_cam(4)
kfs = make_keyframes(5, cam=cam, seed=41)
ref, ora = T.RefChisel(0.05, cam), ora_lib.chisel(0.05)
for kf in kfs:
    ref.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
    ora.integrate(kf["xyz"], kf["rgb"], kf["kfid"], kf["Twc"])
ir = {tuple(int(v) for v in c) for c in ref.chunk_ids()}
io = {tuple(int(v) for v in c) for c in ora.chunk_ids()}
print("chunks: fast-math-contract build", len(ir), "oracle", len(io), "only one side", len(ir ^ io))
nv = nd = 0; mx = 0.0; wd = 0; kd = 0; cd = 0
for cid in ir & io:
    a, b = ref.get_chunk(*cid), ora.get_chunk(*cid)
    known = (a[1] > 0) | (b[1] > 0)
    nv += int(known.sum())
    d = np.abs(a[0][known] - b[0][known])
    nd += int((a[0][known].view(np.uint32) != b[0][known].view(np.uint32)).sum())
    mx = max(mx, float(d[np.isfinite(d) & (d < 1000)].max(initial=0)))
    wd += int((a[1] != b[1]).sum()); kd += int((a[2] != b[2]).sum()); cd += int((a[3] != b[3]).sum())
print("known voxels", nv, "sdf bits differ", nd, "max |dsdf| m", mx, "weight differs", wd, "kfid differs", kd, "colour differs", cd)
