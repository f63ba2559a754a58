"""Loads plvs_amd/synth_scene.py WITHOUT importing the plvs_amd package (whose
import requires the HIP library), so CPU-only tests can generate inputs."""
import importlib.util
import os

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plvs_amd", "synth_scene.py")
_spec = importlib.util.spec_from_file_location("plvs_amd_synth_scene", _p)
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
make_keyframes = _m.make_keyframes
make_rgbd_frames = _m.make_rgbd_frames
TUM1 = _m.TUM1
KITTI = _m.KITTI
