"""The C++ port (second checker of the GPU tests) against the long-horizon goldens of the torch oracle."""
import pytest

from . import long_horizon


@pytest.mark.parametrize("lattice", ["quads", "kagome", "quads32", "quads128"])
def test_cpu_port_follows_the_oracle_over_a_long_horizon(cpu_lib, lattice):
    long_horizon.check(cpu_lib, lattice)


def test_cpu_port_runs_the_pulse_rs_script_as_the_oracle_does(cpu_lib):
    from . import pulse_rs
    pulse_rs.check(cpu_lib)
