"""Cotangent of ``timepoints`` (``DynamicSolver.timepoints_vjp``): ``jax.grad`` through the reference's ``solve_dynamics`` reaches its
``timepoints`` argument (dynamics.py:138-148; jax.experimental.ode._odeint_rev).  Checked on the CPU port against
  (a) the oracle's restatement of _odeint_rev (oracle/ref_adjoint.py: the ts_bar the reference would return), fed by the engine's hooks, and
  (b) central differences of the engine's own solve under a shifted output time (covers the prescribed DOFs' outputs too)."""
import numpy as np
import pytest

from oracle import ref_adjoint as RA
from oracle import ref_ode

from . import adjoint_semantics as AS
from .common import Case

FAST = dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5)
TS_OF = {"quads": np.array([0.0, 1.1e-4, 2.0e-4, 3.2e-4]),
         # (the 3 x 3-cell kagome with this drive turns violent after 2e-4 s: 200 steps per interval no longer resolve it -- halving the step
         # moves the last output by 1e-3 -- and differences of the DISCRETE solve w.r.t. the grid mean nothing there)
         "kagome": np.array([0.0, 0.6e-4, 1.1e-4, 1.7e-4])}
SPI = 200          # fine enough that the derivative of the discrete map and that of the continuous solution agree to ~1e-7


def _solve(c, ts, fb=None):
    nb = c.geo.n_blocks
    fields = c.solver(np.zeros((2, nb, 3)), ts, c.cp, keep_trajectory=fb is not None, steps_per_interval=SPI)
    return np.asarray(fields)


@pytest.mark.parametrize("lattice,contact", [("quads", False), ("kagome", False), ("quads", True), ("kagome", True)])
def test_timepoints_cotangent(cpu_lib, lattice, contact):
    c = Case(lattice, 3, True, contact, seed=5, cutoff_deg=42.0 if lattice == "quads" else 125.0, lib=cpu_lib)
    c.cp = c.cp._replace(constraint_params=FAST)
    TS = TS_OF[lattice]
    nb = c.geo.n_blocks
    rng = np.random.default_rng(11)
    fields = _solve(c, TS, fb=True)
    fb = rng.normal(size=fields.shape) * np.array([1.0, 1e-4])[None, :, None, None]      # displacements ~1e-1, velocities ~1e3
    trees, s0 = c.solver.vjp(fb)
    tsb = c.solver.timepoints_vjp(fb)
    assert tsb.shape == TS.shape and np.all(np.isfinite(tsb)) and np.abs(tsb[1:]).min() > 0

    # (b) central differences of L = sum(fb * fields(timepoints)); shifting t_0 moves the start of the integration (state0 stays at rest).
    # Without contact only: an engaged contact is a kink of the right-hand side, the truncation error of a fixed grid then depends on where
    # the kink falls inside a step, and the derivative of the DISCRETE map w.r.t. an output time picks up that saw-tooth through every
    # later output (measured: a term of the size of the answer that changes sign with the difference step).  The reference's reverse
    # pass -- and timepoints_vjp -- return the derivative of the continuous solution, which has no such term: check (a).
    def loss(ts):
        return float((fb * _solve(c, ts)).sum())
    for i in range(0 if contact else len(TS)):
        eps = 2e-9
        tp, tm = TS.copy(), TS.copy()
        tp[i] += eps; tm[i] -= eps
        fd = (loss(tp) - loss(tm)) / (2 * eps)
        # (entries are sums of large terms of both signs -- ts_bar[0] = -lambda . f in particular -- and the pulse's onset at the input
        # delay is only C1: the derivative of the discrete map and the continuous formula agree to ~1e-5 of the largest entry at this step size)
        assert abs(tsb[i] - fd) <= 2e-5 * np.abs(tsb).max(), (i, tsb[i], fd)

    # (a) what the reference's reverse pass returns for the free-DOF ODE (cotangent on the free DOFs only: the prescribed DOFs are not
    # part of odeint's state there): adaptive Dormand-Prince forward, _odeint_rev backwards, tight tolerances
    prov = AS.EngineRHS(c.solver, c.cp)
    free = prov.free
    fbf = np.zeros_like(fb).reshape(len(TS), 2, nb * 3)
    fbf[:, :, free] = fb.reshape(len(TS), 2, nb * 3)[:, :, free]
    fields = _solve(c, TS, fb=True)
    c.solver.vjp(fbf.reshape(fb.shape))
    tsb_free = c.solver.timepoints_vjp(fbf.reshape(fb.shape))
    ys = ref_ode.odeint(prov.func, np.zeros(2 * prov.n_free), TS, rtol=1e-11, atol=1e-11)
    g = fbf[:, :, free].reshape(len(TS), -1)
    _, ts_ref, _ = RA.odeint_rev(prov.func, prov.vjp, ys, TS, g, prov.args_size, rtol=1e-10, atol=1e-10)
    assert np.abs(tsb_free - ts_ref).max() <= 1e-5 * np.abs(ts_ref).max(), (tsb_free, ts_ref)


def test_timepoints_cotangent_needs_the_reverse_pass_first(cpu_lib):
    c = Case("quads", 3, True, False, seed=1, lib=cpu_lib)
    c.cp = c.cp._replace(constraint_params=FAST)
    f = _solve(c, TS_OF["quads"], fb=True)
    with pytest.raises(RuntimeError):
        c.solver.timepoints_vjp(np.ones_like(f))


@pytest.mark.gpu
@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_timepoints_cotangent_hip(hip_lib, cpu_lib, lattice):
    """The same quantity through the HIP engine (its dfx_rhs hook and its reverse sweep's state0 cotangent) against the CPU port, and
    against central differences of the HIP solve (no contact: see above)."""
    TS = TS_OF[lattice]
    out = {}
    for name, lib, contact in (("hip", None, True), ("cpu", cpu_lib, True), ("hip_nc", None, False)):
        c = Case(lattice, 3, True, contact, seed=5, cutoff_deg=42.0 if lattice == "quads" else 125.0, lib=lib)
        c.cp = c.cp._replace(constraint_params=FAST)
        fields = _solve(c, TS, fb=True)
        fb = np.random.default_rng(11).normal(size=fields.shape) * np.array([1.0, 1e-4])[None, :, None, None]
        c.solver.vjp(fb)
        out[name] = c.solver.timepoints_vjp(fb)
        if name == "hip_nc":
            for i in range(len(TS)):
                tp, tm = TS.copy(), TS.copy()
                tp[i] += 2e-9; tm[i] -= 2e-9
                fd = float((fb * (_solve(c, tp) - _solve(c, tm))).sum()) / 4e-9
                assert abs(out[name][i] - fd) <= 2e-5 * np.abs(out[name]).max(), (i, out[name][i], fd)
    assert np.abs(out["hip"] - out["cpu"]).max() <= 1e-9 * np.abs(out["cpu"]).max()
