#!/usr/bin/env python3
"""Golden digests of the LSD line path made by the REFERENCE ITSELF: Thirdparty/line_descriptor's lsd_custom.cpp,
LSDDetector_custom.cpp and binary_descriptor_custom.cpp and src/LineExtractor.cc compiled here, unmodified, against the OpenCV
stand-in (oracle/ref/Makefile, lsd_ref_wrap.cpp -> oracle/_ref/liblsd_ref.so), run over tests/lsd_golden_scenario.py; the digests
go to tests/golden/lsd_reference_digests.json.  Dev-time tool (needs the compiled reference); tests/test_lsd.py checks the
product's host stages (CPU) and the HIP path (GPU) against the file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import lsd_golden_scenario as S      # noqa: E402
from tests.test_lsd import RefBackend           # noqa: E402


def main():
    out = dict(what="sha1 digests of the cv::Vec4f segments of LineSegmentDetector::detect, of the 68-byte KeyLine records of "
                    "LSDDetectorC::detect, and of the KeyLines + 32-byte LBD descriptors of LineExtractor::operator() with "
                    "skUseLsdExtractor, all produced by the reference's own sources (see this script)",
               cases=S.run(RefBackend()))
    path = os.path.join(ROOT, "tests", "golden", "lsd_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, {k: {i: r["n"] for i, r in v.items()} for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
