"""-m gpu: the HIP engine against the committed golden fixtures and the reference's tensile known-answer test."""
import os

import numpy as np
import pytest

from . import long_horizon
from .common import Case, relerr, tensile_solver

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_rhs(hip_lib):
    gold = np.load(os.path.join(GOLD, "rhs.npz"))
    for lattice, n, cut in (("quads", 4, 42.0), ("kagome", 3, 125.0)):
        for nonlinear in (True, False):
            key = f"{lattice}{n}_{'nl' if nonlinear else 'lin'}"
            c = Case(lattice, n, nonlinear, True, seed=7, lib=None, cutoff_deg=cut)
            y, free = gold[key + "_y"], gold[key + "_free"]
            flat = c.solver._flatten(c.cp)
            c.solver.engine.set_params(**{k: v[None] for k, v in flat.items()})
            dy = c.solver.engine.rhs(y[None], 0.012)[0].reshape(2, -1)[:, free]
            assert relerr(dy, gold[key + "_dy_free"]) < 1e-12


def test_fixed_grid_converges_to_adaptive_reference(hip_lib):
    gold = np.load(os.path.join(GOLD, "adaptive_8x8.npz"))
    c = Case("quads", 8, False, False, damping=False, seed=1, lib=None)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=float(gold["loading_rate"]), input_delay=float(gold["input_delay"])))
    f = c.solver(np.zeros((2, 64, 3)), gold["timepoints"], cp, steps_per_interval=200)
    free = c.solver.free_DOF_ids
    a = f.reshape(len(f), 2, -1)[:, :, free]
    b = gold["fields"].reshape(len(f), 2, -1)[:, :, free]
    assert relerr(a[:, 0], b[:, 0]) < 1e-6 and relerr(a[:, 1], b[:, 1]) < 1e-6   # stated tolerance vs the adaptive reference


def test_golden_focusing_gradient(hip_lib):
    gold = np.load(os.path.join(GOLD, "focusing_6x6.npz"))
    c = Case("quads", 6, True, True, seed=9, lib=None, cutoff_deg=42.0)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    c.solver(np.zeros((2, 36, 3)), gold["timepoints"], cp, keep_trajectory=True, steps_per_interval=int(gold["spi"]))
    obj, tree, _ = c.solver.kinetic_energy_value_and_vjp(gold["target"].astype(np.int32))
    gh, gv = c.geo.vjp((gold["design_h"], gold["design_v"]), tree.geometrical_params.centroid_node_vectors,
                       tree.geometrical_params.block_centroids)
    assert abs(obj - float(gold["objective"])) / float(gold["objective"]) < 1e-10
    assert relerr(gh, gold["grad_h"]) < 1e-9 and relerr(gv, gold["grad_v"]) < 1e-9


@pytest.mark.parametrize("lattice", ["quads", "kagome", "quads32", "quads128"])
def test_long_horizon_trajectory_and_gradient_against_the_oracle(hip_lib, lattice):
    """1200 (quads 8x8) / 800 (kagome 4x4) / 2000 (quads 32x32: several workgroups, several segments; by default the persistent stage
    loop, whose ring wraps 1 500 times) / 500 (quads 128x128: BASELINE config 3's lattice at FULL SIZE, 1 024 waves, every XCD band) steps
    with the contact engaged throughout: trajectory, objective and design gradient of the HIP engine against the torch oracle's taped solve
    (tests/golden/long_horizon_*.npz) -- an oracle that shares no header with the product."""
    long_horizon.check(None, lattice)


@pytest.mark.parametrize("lattice", ["quads32", "quads128"])
def test_long_horizon_with_one_launch_per_stage(hip_lib, monkeypatch, lattice):
    """The same goldens through the stage launches (DFX_PERSIST=0): both forms of the stage loop are pinned to the oracle."""
    monkeypatch.setenv("DFX_PERSIST", "0")
    long_horizon.check(None, lattice)


def test_pulse_rs_script_as_written(hip_lib):
    """scripts/pulse_RS.py: unconstrained 40 x 20 rotated squares, force pulse, the default adaptive call -- against the oracle's golden."""
    from . import pulse_rs
    pulse_rs.check(None)


@pytest.mark.parametrize("n1_cells,strain,nonlinear", [(5, 0.2, False), (5, 0.6, True), (20, 0.4, True), (10, 0.6, False)])
def test_tensile_known_answer(hip_lib, n1_cells, strain, nonlinear):
    """reference tests/test_difflexmm.py:35-146 through the HIP engine."""
    got, _ = tensile_solver(None, n1_cells, strain, nonlinear)
    assert abs(got - strain) / strain < 1e-4


def test_adaptive_matches_reference_odeint_golden(hip_lib):
    """Default call (no steps_per_interval): adaptive Dopri5 with the reference's controller on the device, against the
    golden trajectory of the oracle's restatement of jax.experimental.ode.odeint (rtol = atol = 1e-8)."""
    gold = np.load(os.path.join(GOLD, "adaptive_8x8.npz"))
    c = Case("quads", 8, False, False, damping=False, seed=1, lib=None)
    c.solver.rtol = c.solver.atol = 1e-8
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=float(gold["loading_rate"]), input_delay=float(gold["input_delay"])))
    f = c.solver(np.zeros((2, 64, 3)), gold["timepoints"], cp)
    assert c.solver.stats["step_control"] == "adaptive"
    # same controller, same tableau: the step sequences coincide unless an accept/reject decision sits within rounding of 1
    assert abs(c.solver.stats["steps"] - int(gold["accepted"])) <= 2
    assert relerr(f, gold["fields"]) < 1e-9


def test_adaptive_members_control_their_own_step(hip_lib):
    ts = np.linspace(0.0, 2e-3, 5)
    singles, cps = [], []
    for seed, rate in ((31, 900.0), (32, 2500.0)):
        c = Case("quads", 5, True, True, seed=seed, lib=None, cutoff_deg=42.0)
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=rate, input_delay=1e-5))
        c.solver.rtol = c.solver.atol = 1e-9
        singles.append((c.solver(np.zeros((2, 25, 3)), ts, cp), c.solver.stats["steps"]))
        cps.append(cp)
    cb = Case("quads", 5, True, True, seed=31, lib=None, cutoff_deg=42.0, batch=2)
    cb.solver.rtol = cb.solver.atol = 1e-9
    fb = cb.solver(np.zeros((2, 25, 3)), ts, cps)
    assert singles[0][1] != singles[1][1]          # the two members really take different numbers of steps
    for m in range(2):
        assert np.array_equal(fb[m], singles[m][0])


def test_adaptive_graph_of_attempts_is_reused_and_rebuilt(hip_lib):
    """Small lattices replay a graph of 32 adaptive attempts kept in the handle: a second call with the same arguments reuses it, a
    call with another number of timepoints or another tolerance (both baked into the graph's kernel arguments) rebuilds it -- every
    result must equal the one of a fresh handle bit for bit."""
    def fresh(ts, tol):
        c = Case("quads", 5, True, True, seed=31, lib=None, cutoff_deg=42.0)
        c.solver.rtol = c.solver.atol = tol
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=900.0, input_delay=1e-5))
        return c, cp, c.solver(np.zeros((2, 25, 3)), ts, cp)
    ts5, ts3 = np.linspace(0.0, 2e-3, 5), np.linspace(0.0, 1e-3, 3)
    c, cp, f5 = fresh(ts5, 1e-9)
    assert np.array_equal(c.solver(np.zeros((2, 25, 3)), ts5, cp), f5)            # reused
    f3 = c.solver(np.zeros((2, 25, 3)), ts3, cp)                                  # other T: rebuilt
    assert np.array_equal(f3, fresh(ts3, 1e-9)[2])
    c.solver.rtol = c.solver.atol = 1e-7                                          # other tolerance: rebuilt
    assert np.array_equal(c.solver(np.zeros((2, 25, 3)), ts5, cp), fresh(ts5, 1e-7)[2])
    c.solver.rtol = c.solver.atol = 1e-9
    assert np.array_equal(c.solver(np.zeros((2, 25, 3)), ts5, cp), f5)


def test_energy_splitting_notebook_ratio_on_hip(hip_lib):
    """The reference-computed anchor (tests/notebook_kat.py: notebooks/quads_energy_splitting_3dp_pla_shims.ipynb cell 23, entry 0 =
    0.33490634) through the problem layer on the HIP engine: device-side adaptive Dopri5 at rtol 1e-8 / atol 1e-4, contact, damping,
    pulse drive, kinetic energies of the two targets."""
    from . import notebook_kat as NK
    r, vals = NK.engine_ratio(None)
    assert abs(r - NK.NOTEBOOK_RATIO) < NK.TOL, r
    assert 1.70 < vals[0] < 1.74 and 5.10 < vals[1] < 5.17, vals
    for angle, printed in NK.ANCHORS:       # both printed anchors (cells 23 and 33)
        r, _ = NK.engine_ratio(None, angle)
        assert abs(r - printed) < NK.TOL, (angle, r)
