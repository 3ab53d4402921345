"""linear_mode_analysis on the CPU port (tests/modes_common.py; the HIP run is tests/test_gpu_modes.py)."""
import pytest

from . import modes_common


@pytest.mark.parametrize("lattice,n,contact", [("quads", 4, False), ("quads", 4, True), ("kagome", 3, True)])
def test_linear_mode_analysis_cpu_port(cpu_lib, lattice, n, contact):
    modes_common.check(cpu_lib, lattice, n, contact)
