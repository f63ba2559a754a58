#!/usr/bin/env python3
"""Golden digests of what the REFERENCE's own search functions return — src/ORBmatcher.cc and src/LineMatcher.cc compiled
unmodified from /root/reference (oracle/ref/Makefile -> oracle/_ref/libmatchers_ref.so) — on the cases of
tests/matchers_golden_scenario.py; written to tests/golden/matchers_reference_digests.json.  Dev-time tool."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import matchers_golden_scenario as S  # noqa: E402


def main():
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmatchers_ref.so"))
    out = dict(what="sha1 digests of the match vectors the reference's own ORBmatcher.cc / LineMatcher.cc return (see this script)",
               cases=S.run("ref", ref=ref))
    path = os.path.join(ROOT, "tests", "golden", "matchers_reference_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, [(c["case"][0][:2], c["matches"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
