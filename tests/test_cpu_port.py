"""No-GPU suite: host logic (ControlParams flattening, geometry VJPs, time functions, gradient tree) and the
per-ligament physics shared with the HIP kernels, exercised through the CPU port and checked against the
torch-autograd oracle."""
import numpy as np
import pytest

from . import parity


@pytest.mark.parametrize("lattice,n", [("quads", 4), ("kagome", 3)])
@pytest.mark.parametrize("nonlinear", [True, False])
@pytest.mark.parametrize("contact", [False, True])
def test_rhs_and_vjp_match_autograd(cpu_lib, lattice, n, nonlinear, contact):
    parity.check_rhs_and_vjp(cpu_lib, lattice, n, nonlinear, contact)


@pytest.mark.parametrize("lattice,n,integrator", [("quads", 4, "dopri5"), ("kagome", 3, "rk4"), ("quads", 3, "rk4")])
def test_trajectory_and_discrete_adjoint(cpu_lib, lattice, n, integrator):
    parity.check_trajectory_and_adjoint(cpu_lib, lattice, n, integrator)


def test_linearized_no_contact_adjoint(cpu_lib):
    parity.check_trajectory_and_adjoint(cpu_lib, "quads", 4, "dopri5", nonlinear=False, contact=False)
