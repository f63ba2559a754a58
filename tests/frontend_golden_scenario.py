"""The front-end golden scenario: ORB key points / descriptors and EDLines KeyLines / LBD descriptors of fixed images
under the shipped parameter sets.  Run three ways over the SAME code below:
  scripts/make_frontend_golden.py   the reference's own sources compiled here (oracle/_ref/libfrontend_ref.so)
                                    -> tests/golden/frontend_reference_digests.json
  tests/test_oracle_pinned_frontend.py, CPU   oracle/orb.cpp, oracle/lines.cpp reproduce the file
  tests/test_oracle_pinned_frontend.py, GPU   the HIP path, through the C ABI, reproduces the file
An extractor pair is anything with  orb(nfeatures) -> f(img, lap) = (mono, kps, desc)  and
lines(nfeatures, shared_pyramid_from=None) -> f(img) = (keylines, desc)."""
import hashlib

import numpy as np

from tests.oracle_lib import golden
from tests.test_orb import synth_frame

IMAGES = ["aloe_640x480.pgm", "aloe_640x480_shift.pgm", "cones_640x480.pgm", "urban1_1241x376.pgm", "synth3", "synth7"]
ORB_FEATURES = [1000, 2000]
LAPPING = {"urban1_1241x376.pgm": (400, 800)}    # a lapping area (fisheye stereo packing) on one image


def image(name):
    return synth_frame(int(name[5:])) if name.startswith("synth") else golden(name)


def _sha(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def orb_case(extract, img, lap):
    mono, kps, desc = extract(img, lap)
    return dict(n=int(len(kps)), mono=int(mono), keypoints=_sha(kps), descriptors=_sha(desc))


def lines_case(extract, img):
    kl, desc = extract(img)
    return dict(n=int(len(kl)), keylines=_sha(kl), descriptors=_sha(desc))


def run(make_orb, make_lines, make_shared):
    """make_orb(nfeatures) -> extract(img, lap); make_lines(nfeatures) -> extract(img);
    make_shared(img) -> (keylines, desc) of the line extractor fed the ORB pyramid (Line.pyramidPrecomputation)."""
    out = {}
    for name in IMAGES:
        img = image(name)
        lap = LAPPING.get(name, (0, 0))
        rec = {}
        for nf in ORB_FEATURES:
            rec[f"orb{nf}"] = orb_case(make_orb(nf), img, lap)
        rec["lines100"] = lines_case(make_lines(100), img)
        rec["lines_all"] = lines_case(make_lines(0), img)
        kl, desc = make_shared(img)
        rec["lines100_shared_pyramid"] = dict(n=int(len(kl)), keylines=_sha(kl), descriptors=_sha(desc))
        out[name] = rec
    return out
