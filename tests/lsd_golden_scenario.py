"""The LSD golden scenario (Line.LSD.on: 1).  Run three ways over the SAME code below:
  scripts/make_lsd_golden.py     the reference's own lsd_custom.cpp / LSDDetector_custom.cpp / binary_descriptor_custom.cpp /
                                 LineExtractor.cc compiled here (oracle/_ref/liblsd_ref.so) -> tests/golden/lsd_reference_digests.json
  tests/test_lsd.py, CPU         the product's host stages (plvs_amd/csrc/lsd_host.hpp, compiled by g++ behind plain-loop
                                 restatements of the three device kernels: tests/host/lsd_host.cpp) reproduce the `segments` part
  tests/test_lsd.py, GPU         the HIP path, through the C ABI, reproduces the whole file
A backend offers  segments(image, refine, scale, sigma_scale, quant, ang_th, log_eps, density_th, n_bins) -> [n, 4] f32
                  detect(image, num_octaves, pyramid_scale, opts (dict of the eight above), min_length) -> keylines (68-byte records)
                  extract(image, nfeatures, num_octaves, opts, min_length) -> (keylines, descriptors [n, 32] u8);
a backend without detect / extract runs the `segments` cases only."""
import hashlib

import numpy as np

from tests.oracle_lib import golden

DEFAULTS = dict(refine=2, scale=0.8, sigma_scale=0.6, quant=2.0, ang_th=22.5, log_eps=0.0, density_th=0.7, n_bins=1024)
# what Tracking's settings parser hands the extractor when the YAML only says Line.LSD.on: 1 (src/Tracking.cc:1466-1485);
# Line.scaleFactor 1.2 is read into a float
TRACKING = dict(refine=1, scale=float(np.float32(1.2)), sigma_scale=0.6, quant=2.0, ang_th=22.5, log_eps=1.0, density_th=0.6,
                n_bins=1024)

IMAGES = ["aloe_640x480.pgm", "cones_640x480.pgm", "urban1_1241x376.pgm"]

SEGMENT_CASES = [
    dict(id="defaults_adv_0.8", image="aloe_640x480.pgm", opts={}),
    dict(id="std_0.8", image="cones_640x480.pgm", opts=dict(refine=1)),
    dict(id="none_0.8", image="urban1_1241x376.pgm", opts=dict(refine=0)),
    dict(id="tracking_1.2", image="urban1_1241x376.pgm", opts=TRACKING),
    dict(id="tracking_1.2_aloe", image="aloe_640x480.pgm", opts=TRACKING),
    dict(id="unscaled_adv", image="cones_640x480.pgm", opts=dict(scale=1.0)),
    dict(id="half_fine_bins", image="aloe_640x480.pgm", opts=dict(scale=0.5, n_bins=256, ang_th=30.0, quant=1.0)),
    dict(id="magnified_sqrt2", image="cones_640x480.pgm", opts=dict(scale=float(np.float32(np.sqrt(2.0))), refine=1, density_th=0.5)),
    dict(id="odd_crop", image="urban1_1241x376.pgm", crop=(3, 5, 517, 301), opts=dict(refine=2, log_eps=1.0)),
]
DETECT_CASES = [
    dict(id="three_octaves_1.2", image="aloe_640x480.pgm", num_octaves=3, pyramid_scale=1.2, opts=TRACKING, min_length=0.025),
    dict(id="two_octaves_sqrt2_adv", image="urban1_1241x376.pgm", num_octaves=2, pyramid_scale=float(np.float32(np.sqrt(2.0))),
         opts=dict(scale=0.8), min_length=0.01),
    dict(id="one_octave_defaults", image="cones_640x480.pgm", num_octaves=1, pyramid_scale=2.0, opts={}, min_length=0.025),
]
EXTRACT_CASES = [
    dict(id="tum_100", image="aloe_640x480.pgm", nfeatures=100, num_octaves=3, opts=TRACKING, min_length=0.025),
    dict(id="kitti_100", image="urban1_1241x376.pgm", nfeatures=100, num_octaves=3, opts=TRACKING, min_length=0.025),
    dict(id="all_lines_two_octaves", image="cones_640x480.pgm", nfeatures=0, num_octaves=2, opts=TRACKING, min_length=0.02),
    dict(id="adv_sqrt2_200", image="cones_640x480.pgm", nfeatures=200, num_octaves=3,
         opts=dict(TRACKING, refine=2, scale=float(np.float32(np.sqrt(2.0)))), min_length=0.025),
]


def image_of(case):
    img = golden(case["image"])
    if "crop" in case:
        x, y, w, h = case["crop"]
        img = img[y:y + h, x:x + w]
    return np.ascontiguousarray(img)


def options(case):
    return dict(DEFAULTS, **case["opts"])


def _sha(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(backend, parts=("segments", "detect", "extract")):
    out = {}
    if "segments" in parts:
        out["segments"] = {}
        for c in SEGMENT_CASES:
            s = backend.segments(image_of(c), **options(c))
            out["segments"][c["id"]] = dict(n=int(len(s)), sha=_sha(s))
    if "detect" in parts:
        out["detect"] = {}
        for c in DETECT_CASES:
            kl = backend.detect(image_of(c), c["num_octaves"], c["pyramid_scale"], options(c), c["min_length"])
            out["detect"][c["id"]] = dict(n=int(len(kl)), sha=_sha(kl))
    if "extract" in parts:
        out["extract"] = {}
        for c in EXTRACT_CASES:
            kl, d = backend.extract(image_of(c), c["nfeatures"], c["num_octaves"], options(c), c["min_length"])
            out["extract"][c["id"]] = dict(n=int(len(kl)), keylines=_sha(kl), descriptors=_sha(d))
    return out
