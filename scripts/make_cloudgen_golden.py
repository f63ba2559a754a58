#!/usr/bin/env python3
"""Golden digests of the depth image -> cloud step made by the REFERENCE ITSELF: PointCloudMapping::InitCamGridPoints and
::GeneratePointCloudInCameraFrameBGRA, cut verbatim out of /root/reference/src/PointCloudMapping.cc and compiled here against
stand-ins (oracle/ref/Makefile, cloudgen_ref_wrap.cpp -> oracle/_ref/libcloudgen_ref.so), run
tests/cloudgen_golden_scenario.py; the digests go to tests/golden/cloudgen_reference_digests.json.  Dev-time tool (needs the
compiled reference); tests/test_oracle_pinned_cloudgen.py checks the oracle (CPU) and the HIP path (GPU) against the file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cloudgen_golden_scenario as S                 # noqa: E402
from tests.test_oracle_pinned_cloudgen import ref_generator     # noqa: E402


def main():
    out = dict(what="sha1 digests of matCamGridPoints_, the cloud's points (48-byte pcl::PointSurfelSegment records) and "
                    "pixelToPointIndex produced by the reference's own InitCamGridPoints / GeneratePointCloudInCameraFrameBGRA "
                    "(see this script)", cases=S.run(ref_generator))
    path = os.path.join(ROOT, "tests", "golden", "cloudgen_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, {k: v["frame0"]["n"] for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
