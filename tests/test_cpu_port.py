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


def test_grid_with_its_own_step_count_per_interval(cpu_lib):
    """dfx_forward_grid: forward + discrete adjoint on a grid with a different number of steps in every output
    interval, against the oracle's unrolled autograd on the same grid."""
    parity.check_trajectory_and_adjoint(cpu_lib, "quads", 4, "dopri5", spi=np.array([3, 7, 1, 5]))


@pytest.mark.parametrize("integrator", ["dopri5", "rk4"])
def test_grid_with_caller_chosen_step_boundaries(cpu_lib, integrator):
    """dfx_forward_grid(step_times=...): unequal steps inside the intervals."""
    parity.check_trajectory_and_adjoint(cpu_lib, "quads", 4, integrator, spi=np.array([3, 7, 2, 5]), own_step_times=True)


def test_step_times_are_validated(cpu_lib):
    from .common import Case
    c = Case("quads", 3, True, False, seed=1, lib=cpu_lib)
    ts = np.array([0.0, 1e-4, 2e-4])
    y0 = np.zeros((2, 9, 3))
    with pytest.raises(RuntimeError, match="strictly increasing"):
        c.solver(y0, ts, c.cp, steps_per_interval=[2, 1], step_times=[0.0, 0.5e-4, 0.5e-4, 2e-4])
    with pytest.raises(RuntimeError, match="every timepoint"):
        c.solver(y0, ts, c.cp, steps_per_interval=[2, 1], step_times=[0.0, 0.5e-4, 1.1e-4, 2e-4])


@pytest.mark.parametrize("kw", [dict(), dict(lattice="kagome", n=3, n_out=121), dict(batch=2, contact=False, nonlinear=False)],
                         ids=["quads-contact", "kagome-contact", "linearized-batch2"])
def test_adaptive_solve_is_differentiable_as_it_stands(cpu_lib, kw):
    """keep_trajectory=True without a grid (the reference's default call under jax.grad, dynamics.py:166): ONE adaptive pass that keeps its
    accepted steps + the dense-output discrete adjoint, against autograd through the oracle's replay of the same steps."""
    parity.check_adaptive_records_adjoint(cpu_lib, **kw)


def test_adaptive_grid_makes_the_default_solve_differentiable(cpu_lib, monkeypatch):
    """keep_trajectory=True without a grid, the two-pass form (DFX_ADAPTIVE_RECORDS=0; the only form before round 6, and what
    ``grid_refine > 1`` still uses): the step boundaries the adaptive controller accepted (plus the output times)
    become the fixed grid (dfx_adaptive_step_times -> dfx_forward_grid); the result stays within the controller's
    tolerance of the adaptive solve and vjp is the exact gradient of the frozen-grid solve (finite differences of it)."""
    from .common import Case
    c = Case("quads", 4, True, True, seed=4, lib=cpu_lib, cutoff_deg=42.0)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    ts = np.array([0.0, 0.5e-4, 1.0e-4, 2.5e-4, 3.0e-4])
    y0 = c.random_state(0.05, 0.02, 5.0)
    s = c.solver
    monkeypatch.setenv("DFX_ADAPTIVE_RECORDS", "0")
    adaptive = s(y0, ts, cp)
    counts = s.engine.adaptive_step_counts()
    times = s.engine.adaptive_step_times(0)
    assert counts.shape == (1, len(ts) - 1) and counts.sum() == s.stats["steps"] == len(times)
    assert np.all(np.diff(times) > 0) and times[-1] >= ts[-1]
    frozen = s(y0, ts, cp, keep_trajectory=True)
    assert s.stats["step_control"] == "adaptive-grid"
    grid, spis = s.stats["step_times"], np.asarray(s.stats["steps_per_interval"])
    assert spis.sum() + 1 == len(grid) <= counts.sum() + len(ts) and np.all(np.isin(ts, grid))
    scale = np.abs(adaptive).max()
    assert np.abs(frozen - adaptive).max() < 2e-6 * scale      # same steps; dense output vs landing on the output times
    fb = np.random.default_rng(1).normal(size=frozen.shape)
    tree, _ = s.vjp(fb)
    g = tree.constraint_params["amplitude"]
    eps = 1e-6
    kw = dict(steps_per_interval=spis, step_times=grid)
    lp = (fb * s(y0, ts, cp._replace(constraint_params=dict(cp.constraint_params, amplitude=7.5 + eps)), **kw)).sum()
    lm = (fb * s(y0, ts, cp._replace(constraint_params=dict(cp.constraint_params, amplitude=7.5 - eps)), **kw)).sum()
    fd = (lp - lm) / (2 * eps)
    assert abs(g - fd) < 2e-6 * abs(fd), (g, fd)


def test_recorded_signal_as_prescribed_displacement(cpu_lib):
    parity.check_table_drive(cpu_lib)


def test_several_dofs_of_one_block_share_a_time_function(cpu_lib):
    parity.check_several_dofs_of_one_block_share_a_time_function(cpu_lib)
