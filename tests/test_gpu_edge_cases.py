"""-m gpu: the edge cases of tests/test_edge_cases.py on the HIP engine."""
import pytest

from .test_edge_cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_hip(hip_lib, name):
    CASES[name](None)
