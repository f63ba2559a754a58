"""The search functions against digests of what the REFERENCE's own code returned (scripts/make_matchers_golden.py ran
ORBmatcher.cc / LineMatcher.cc, compiled unmodified, on tests/matchers_golden_scenario.py): the oracle on CPU, the HIP
path through the C ABI on the GPU — nothing of the reference is needed at test time."""
import json
import os

import pytest

from tests import matchers_golden_scenario as S
from tests import oracle_lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matchers_reference_digests.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        g = json.load(f)
    assert [c["case"] for c in g["cases"]] == [list(c) for c in S.CASES], "cases changed: regenerate with scripts/make_matchers_golden.py"
    return g["cases"]


def _check(got, golden):
    for a, b in zip(got, golden):
        assert a == b, f"{a['case']}: {a} differs from what the reference's code returned: {b}"
        assert a["matches"] > 5


def test_oracle_reproduces_what_the_reference_search_functions_returned(golden):
    _check(S.run("oracle", oracle=oracle_lib.load()), golden)


@pytest.mark.gpu
def test_hip_reproduces_what_the_reference_search_functions_returned(golden):
    _check(S.run("hip"), golden)
