"""Edge cases of the solver API on the CPU port (host logic + shared physics); the same cases run on the HIP engine in
tests/test_gpu_edge_cases.py.  Reference behaviours cited where they exist."""
import math

import numpy as np
import pytest

import difflexmm_amd as dm
from difflexmm_amd import energy as E
from difflexmm_amd import geometry as G
from difflexmm_amd import loading as L
from difflexmm_amd.dynamics import setup_dynamic_solver

from .common import Case, relerr

CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


def _chain(lib, n=3, **kw):
    g = G.RotatedSquareGeometry(n, 1, spacing=1.0)
    en = E.build_strain_energy(g.bond_connectivity(), E.ligament_energy)
    cp = dm.ControlParams(dm.GeometricalParams(g.block_centroids(0.0), g.centroid_node_vectors(0.0)),
                          dm.MechanicalParams(dm.LigamentParams(1.0, 0.02, 4e-5, g.reference_bond_vectors()), None,
                                              np.tile([1.0, 1.0, 0.076], (g.n_blocks, 1)), kw.pop("damping", 0.0)))
    return g, en, cp


@case
def single_timepoint_returns_reconstructed_initial_state(lib):
    """dynamics.py:138-148: row 0 of the result is the initial state (constrained DOFs overridden by c(t0))."""
    c = Case("quads", 4, True, False, seed=1, lib=lib)
    y0 = c.random_state(0.01, 0.01, 1.0)
    f = c.solver(y0, np.array([0.004]), c.cp)
    assert f.shape == (1, 2, 16, 3)
    free = c.solver.free_DOF_ids
    assert np.array_equal(f[0].reshape(2, -1)[:, free], y0.reshape(2, -1)[:, free])
    con = c.solver.constrained_DOF_ids
    tau = 0.004 - c.cp.constraint_params["input_delay"]
    expect = 7.5 * 0.5 * (1 - math.cos(2 * math.pi * 30 * tau))
    assert abs(f[0, 0].reshape(-1)[con[0]] - expect) < 1e-14 and np.all(f[0, 0].reshape(-1)[con[1:]] == 0)


@case
def one_row_of_per_member_timepoints_returns_the_same_shape_as_a_plain_grid(lib):
    """round-3 advice: a (1, T) array of output times took the per-member branch and came back with a leading member axis that the
    same solve with 1-D timepoints does not have."""
    c = Case("quads", 4, True, False, seed=1, lib=lib)
    ts = np.linspace(0.0, 2e-3, 3)
    a = c.solver(np.zeros((2, 16, 3)), ts, c.cp, steps_per_interval=5)
    b = c.solver(np.zeros((2, 16, 3)), ts[None, :], c.cp, steps_per_interval=5)
    assert a.shape == b.shape == (3, 2, 16, 3) and np.array_equal(a, b)
    assert c.solver(None, ts[None, :], c.cp, steps_per_interval=5).shape == (3, 2, 16, 3)      # state0 = None: at rest


@case
def no_constraints_no_damping_conserves_energy(lib):
    """defaults of dynamics.py:60-69 (no constrained pairs, damped_blocks=None): free vibration conserves E_kin + E_pot."""
    g, en, cp = _chain(lib)
    solver = setup_dynamic_solver(g, en, _lib=lib)
    y0 = np.zeros((2, g.n_blocks, 3))
    y0[1, :, 0] = np.linspace(-0.05, 0.05, g.n_blocks)
    f = solver(y0, np.linspace(0, 20.0, 6), cp, steps_per_interval=400)
    inertia = np.asarray(cp.mechanical_params.inertia)
    flat = solver._flatten(cp)
    solver.engine.set_params(**{k: v[None] for k, v in flat.items()})
    tot = [E.kinetic_energy(f[k, 1], inertia) + solver.engine.energy(f[k, 0][None])[0] for k in range(len(f))]
    assert max(tot) - min(tot) < 1e-9 * tot[0] and tot[0] > 0
    assert abs((f[-1, 1] * inertia).sum(0)[0] - (y0[1] * inertia).sum(0)[0]) < 1e-12   # linear momentum


@case
def scalar_damping_and_subset_of_damped_blocks(lib):
    """loading.py:71-106: damping may be a scalar and act on a subset of blocks."""
    g, en, cp = _chain(lib, damping=0.3)
    damped = np.array([1, 4])
    solver = setup_dynamic_solver(g, en, damped_blocks=damped, _lib=lib)
    flat = solver._flatten(cp)
    assert np.all(flat["damping"][damped] == 0.3) and abs(flat["damping"].sum() - 1.8) < 1e-12
    y0 = np.zeros((2, g.n_blocks, 3)); y0[1, :, 1] = 0.1
    f = solver(y0, np.linspace(0, 5.0, 3), cp, keep_trajectory=True, steps_per_interval=100)
    tree, _ = solver.vjp(np.ones_like(f))
    assert np.ndim(tree.mechanical_params.damping) == 0 and np.isfinite(tree.mechanical_params.damping)
    eps = 1e-6
    lp = solver(y0, np.linspace(0, 5.0, 3), cp._replace(mechanical_params=cp.mechanical_params._replace(damping=0.3 + eps)), steps_per_interval=100).sum()
    lm = solver(y0, np.linspace(0, 5.0, 3), cp._replace(mechanical_params=cp.mechanical_params._replace(damping=0.3 - eps)), steps_per_interval=100).sum()
    assert abs(tree.mechanical_params.damping - (lp - lm) / (2 * eps)) < 1e-5 * abs((lp - lm) / (2 * eps))


@case
def force_loading_sech2tanh_and_its_parameter_gradient(lib):
    """scripts/pulse_RS.py:49-50 force pulse through build_loading (loading.py:12-47) with named loading_params."""
    g, en, cp = _chain(lib, damping=0.05)
    loaded = np.array([[g.n1_blocks - 1, 0], [g.n_blocks - 1, 0]])
    solver = setup_dynamic_solver(g, en, loaded_block_DOF_pairs=loaded, loading_fn=L.Sech2Tanh(amplitude="amp", width="w"),
                                  constrained_block_DOF_pairs=np.array([[0, 0], [0, 1], [g.n1_blocks, 0]]),
                                  damped_blocks=np.arange(g.n_blocks), _lib=lib)
    cp = cp._replace(loading_params=dict(amp=0.02, w=1.5))
    ts = np.linspace(0, 12.0, 4)
    f = solver(np.zeros((2, g.n_blocks, 3)), ts, cp, keep_trajectory=True, steps_per_interval=150)
    assert np.abs(f[-1, 0]).max() > 1e-4
    fb = np.random.default_rng(0).normal(size=f.shape)
    tree, _ = solver.vjp(fb)
    for name, eps in (("amp", 1e-7), ("w", 1e-6)):
        lp = (fb * solver(np.zeros((2, g.n_blocks, 3)), ts, cp._replace(loading_params=dict(cp.loading_params, **{name: cp.loading_params[name] + eps})), steps_per_interval=150)).sum()
        lm = (fb * solver(np.zeros((2, g.n_blocks, 3)), ts, cp._replace(loading_params=dict(cp.loading_params, **{name: cp.loading_params[name] - eps})), steps_per_interval=150)).sum()
        fd = (lp - lm) / (2 * eps)
        assert abs(tree.loading_params[name] - fd) < 2e-5 * abs(fd), name


@case
def harmonic_drive_plus_static_ramp_two_time_functions(lib):
    """Two time functions at once: harmonic displacement drive (problems/quads_spin.py:210-222) and a ramp force."""
    g, en, cp = _chain(lib, damping=0.05)
    con = np.array([[0, 0], [0, 1], [0, 2]])
    solver = setup_dynamic_solver(g, en, loaded_block_DOF_pairs=np.array([[g.n_blocks - 1, 1]]), loading_fn=L.Ramp(amplitude=0.01, rate=0.2),
                                  constrained_block_DOF_pairs=con, constrained_DOFs_fn=L.Harmonic(np.array([1.0, 0.0, 0.0])),
                                  damped_blocks=np.arange(g.n_blocks), _lib=lib)
    cp = cp._replace(constraint_params=dict(amplitude=0.05, loading_rate=0.25, input_delay=1.0))
    ts = np.linspace(0, 9.0, 4)
    f = solver(np.zeros((2, g.n_blocks, 3)), ts, cp, steps_per_interval=120)
    expect = [0.05 * 0.5 * (1 - math.cos(2 * math.pi * 0.25 * (t - 1.0))) if t > 1.0 else 0.0 for t in ts]
    assert np.allclose(f[:, 0, 0, 0], expect, atol=1e-15)
    assert np.abs(f[-1, 0, -1, 1]) > 1e-4
    with pytest.raises(ValueError):
        setup_dynamic_solver(g, en, loaded_block_DOF_pairs=np.array([[1, 1]]), loading_fn=L.Ramp() + L.Constant(),
                             constrained_block_DOF_pairs=con, constrained_DOFs_fn=L.Harmonic(1.0), _lib=lib)


@case
def arbitrary_python_callables_are_rejected(lib):
    g, en, cp = _chain(lib)
    with pytest.raises(TypeError):
        setup_dynamic_solver(g, en, constrained_block_DOF_pairs=np.array([[0, 0]]), constrained_DOFs_fn=lambda t: 0.1 * t, _lib=lib)
    with pytest.raises(TypeError):
        setup_dynamic_solver(g, lambda u, cp: 0.0, _lib=lib)
    assert E.build_contact_energy(g.bond_connectivity(), angle_based=False).spec.contact == 2       # distance-based: built in round 2


@case
def adaptive_default_matches_tight_fixed_grid(lib):
    """Reference default (rtol = atol = 1e-8, dynamics.py:68-69) vs a fine fixed grid: same ODE solution."""
    c = Case("kagome", 3, True, False, seed=4, lib=lib)
    cp = c.cp._replace(constraint_params=dict(amplitude=2.0, loading_rate=800.0, input_delay=1e-4))
    ts = np.linspace(0, 2e-3, 5)
    fa = c.solver(np.zeros((2, 18, 3)), ts, cp)
    ff = c.solver(np.zeros((2, 18, 3)), ts, cp, steps_per_interval=400)
    assert c.solver.stats["step_control"] == "fixed"
    assert relerr(fa[:, 0], ff[:, 0]) < 1e-6 and relerr(fa[:, 1], ff[:, 1]) < 1e-6


@case
def invalid_problems_are_refused(lib):
    g, en, cp = _chain(lib)
    bad = np.array([[0, 6], [0, 10]])         # node 0 with two ligaments: fine since round 3, except with the distance-based contact
    setup_dynamic_solver(g, E.build_strain_energy(bad, E.ligament_energy), _lib=lib)
    with pytest.raises(RuntimeError, match="more than one ligament"):
        setup_dynamic_solver(g, E.combine_block_energies(E.build_strain_energy(bad, E.ligament_energy),
                                                          E.build_contact_energy(bad, angle_based=False)), _lib=lib)
    solver = setup_dynamic_solver(g, en, _lib=lib)
    with pytest.raises(RuntimeError, match="inertia must be positive"):
        solver(np.zeros((2, g.n_blocks, 3)), np.array([0.0, 1.0]),
               cp._replace(mechanical_params=cp.mechanical_params._replace(inertia=np.zeros((g.n_blocks, 3)))), steps_per_interval=2)
    with pytest.raises(RuntimeError, match="keep_trajectory"):
        solver.engine.adjoint(np.zeros((1, 2, 2, g.n_blocks, 3)))


@case
def adaptive_solve_from_rest_without_a_state(lib):
    """state0 == NULL means "at rest" for the adaptive forward too (include/dfx.h; round-2 advice: only the fixed-grid entry
    points handled it)."""
    c = Case("quads", 4, True, False, lib=lib)
    c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    flat = c.solver._flatten(c.cp)
    c.solver.engine.set_params(**{k: v[None] for k, v in flat.items()})
    ts = np.linspace(0.0, 2e-4, 3)
    f0, st0 = c.solver.engine.forward_adaptive(None, ts, 1e-8, 1e-8)
    f1, st1 = c.solver.engine.forward_adaptive(np.zeros((1, 2, 16, 3)), ts, 1e-8, 1e-8)
    assert st0["steps"] == st1["steps"] > 0 and np.array_equal(f0, f1) and np.abs(f0).max() > 0


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_cpu_port(cpu_lib, name):
    CASES[name](cpu_lib)
