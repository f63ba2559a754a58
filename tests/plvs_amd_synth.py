"""The synthetic RGB-D inputs of the tests and of bench.py (tests/synth_scene.py: input generation only, numpy on the
host — kept out of the product package).  This module keeps the names older tests import."""
from tests import synth_scene as _m

make_keyframes = _m.make_keyframes
make_rgbd_frames = _m.make_rgbd_frames
TUM1 = _m.TUM1
KITTI = _m.KITTI
