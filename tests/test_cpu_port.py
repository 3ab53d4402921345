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


def test_cotangents_on_prescribed_dof_outputs_reach_constraint_params(cpu_lib):
    """fields[:, :, driven DOFs] = c(t_k), c'(t_k) depend on constraint_params directly (dynamics.py:132-134,169-182);
    total derivative (dynamics + direct) against central finite differences of the engine's own forward."""
    from .common import Case
    c = Case("quads", 4, True, False, seed=2, lib=cpu_lib)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    ts = np.linspace(0, 3e-4, 5)
    f = c.solver(np.zeros((2, 16, 3)), ts, cp, keep_trajectory=True, steps_per_interval=6)
    fb = np.random.default_rng(0).normal(size=f.shape)
    tree, _ = c.solver.vjp(fb)

    def loss(par, val):
        cp2 = cp._replace(constraint_params=dict(cp.constraint_params, **{par: val}))
        return (fb * c.solver(np.zeros((2, 16, 3)), ts, cp2, steps_per_interval=6)).sum()

    for par, eps, tol in (("amplitude", 1e-5, 1e-5), ("loading_rate", 1e-3, 1e-4)):
        v = cp.constraint_params[par]
        fd = (loss(par, v + eps) - loss(par, v - eps)) / (2 * eps)
        assert abs(tree.constraint_params[par] - fd) / abs(fd) < tol, par
