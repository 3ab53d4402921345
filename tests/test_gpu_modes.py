"""-m gpu: linear_mode_analysis with the stiffness matrix assembled by the HIP engine's Hessian-vector hook."""
import pytest

from . import modes_common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lattice,n,contact", [("quads", 4, False), ("quads", 4, True), ("kagome", 3, True), ("quads", 10, True)])
def test_linear_mode_analysis_hip(hip_lib, lattice, n, contact):
    modes_common.check(None, lattice, n, contact)
