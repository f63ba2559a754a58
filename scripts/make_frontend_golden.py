#!/usr/bin/env python3
"""Golden digests of the front end built by the REFERENCE ITSELF: src/ORBextractor.cc, src/LineExtractor.cc and
Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp compiled unmodified from /root/reference against the OpenCV
stand-in oracle/ref/cv_full (oracle/ref/Makefile -> oracle/_ref/libfrontend_ref.so) run tests/frontend_golden_scenario.py;
the digests go to tests/golden/frontend_reference_digests.json.  Dev-time tool (needs the compiled reference);
tests/test_oracle_pinned_frontend.py checks the oracle (CPU) and the HIP path (GPU) against the committed file.
What the digests pin: everything that is PLVS's own code on this path; the OpenCV primitives underneath (FAST, resize,
GaussianBlur, Sobel, fastAtan2, ...) are the restatements of oracle/cv_primitives.hpp on both sides."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import frontend_golden_scenario as S                    # noqa: E402
from tests.test_oracle_pinned_frontend import ref_extractors       # noqa: E402


def main():
    out = dict(what="sha1 digests of ORB key points / descriptors and KeyLines / LBD descriptors produced by the "
                    "reference's own ORBextractor.cc, LineExtractor.cc and binary_descriptor_custom.cpp (see this script)",
               cases=S.run(*ref_extractors()))
    path = os.path.join(ROOT, "tests", "golden", "frontend_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, {k: (v["orb2000"]["n"], v["lines100"]["n"], v["lines_all"]["n"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
