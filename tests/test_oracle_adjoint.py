"""The oracle's restatement of jax.experimental.ode._odeint_rev (oracle/ref_adjoint.py): the continuous adjoint the reference's
jax.grad computes (difflexmm/dynamics.py:166, problems/quads_focusing.py:565).  No JAX here: it is checked for consistency -- at tight
tolerances it must converge to the exact gradient of the problem, which autograd through a finely discretised solve also converges
to -- and used to put a number on the distance between the reference's own gradient at ITS tolerances and the exact one."""
import numpy as np
import torch

from oracle import ref_adjoint as RA
from oracle import ref_dynamics as OD
from oracle import ref_geometry as OG
from oracle import ref_ode

from .common import DENSITY, Case, relerr

FAST = dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5)
NAMES = ["cen", "cnv", "ks", "ksh", "kr", "refv", "density", "damping", "k_contact", "min_angle", "cutoff_angle",
         "amplitude", "loading_rate", "input_delay", "inertia"]       # the leaves of odeint's (control_params, inertia) arguments


def _setup(cpu_lib, n=3):
    # a SMOOTH problem (no contact): the convergence check below compares two discretisations at 1e-7; with the contact penalty
    # switching on and off (a kink in the force) the fixed grid converges at first order only, the adaptive controller refines there
    return Case("quads", n, True, False, seed=3, lib=cpu_lib)       # only the oracle side of the case is used


def _leaves(c, solver):
    T = lambda x: torch.tensor(np.array(x, dtype=np.float64), requires_grad=True)   # noqa: E731
    vals = dict(cen=c.cen, cnv=c.cnv, ks=c.ks, ksh=c.ksh, kr=c.kr, refv=c.refv, density=DENSITY, damping=c.dval,
                k_contact=c.contact_params[2], min_angle=c.contact_params[0], cutoff_angle=c.contact_params[1], **FAST)
    # (contact parameters stay in the list although this problem has no contact: they are leaves of the reference's ControlParams
    # whenever a problem passes them, take part in the error norm of the reverse integration, and get zero gradients)
    free = torch.as_tensor(solver.free_DOF_ids, dtype=torch.long)
    vals["inertia"] = OG.compute_inertia(torch.as_tensor(np.array(c.cnv)), torch.tensor(DENSITY, dtype=torch.float64)).reshape(-1)[free].numpy()
    return [T(vals[k]) for k in NAMES]


def test_continuous_adjoint_converges_to_the_exact_gradient_and_paper_tolerances_are_measured(cpu_lib):
    c = _setup(cpu_lib)
    solver = c.oracle_solver(integrator="adaptive", rtol=1e-10, atol=1e-10)
    leaves = _leaves(c, solver)
    cp_of = lambda L: c.oracle_cp({k: v for k, v in zip(NAMES, L) if k != "inertia"})      # noqa: E731
    prov = RA.TorchRHS(solver, cp_of, lambda L: L[-1], leaves)
    n_free = prov.n_free
    ts = np.linspace(0.0, 2e-4, 3)
    free = list(solver.free_DOF_ids)
    tdofs = [free.index(4 * 3 + d) for d in range(3)]                # block 4: the free centre block
    m_t = leaves[-1].detach().numpy()[tdofs]

    def objective_bar(ys):           # sum_t m v^2 / 2 on the target block: cotangent of the (T, 2 n_free) solution
        g = np.zeros_like(ys)
        for d, mm in zip(tdofs, m_t):
            g[:, n_free + d] = mm * ys[:, n_free + d]
        return g, float(sum(0.5 * mm * (ys[:, n_free + d] ** 2).sum() for d, mm in zip(tdofs, m_t)))

    def continuous(rtol, atol):
        st = {}
        ys = ref_ode.odeint(prov.func, np.zeros(2 * n_free), ts, rtol=rtol, atol=atol, stats=st)
        g, val = objective_bar(ys)
        st_r = {}
        _, _, args_bar = RA.odeint_rev(prov.func, prov.vjp, ys, ts, g, prov.args_size, rtol=rtol, atol=atol, stats=st_r)
        bars = dict(zip(NAMES, prov.split(args_bar)))
        return val, bars, st, st_r

    # exact gradient: autograd through a fixed grid fine enough for 1e-9 (5th order, 100 steps per interval)
    fsolver = c.oracle_solver(integrator="fixed", steps_per_interval=100)
    hist, _ = OD.solve_fixed_differentiable(fsolver, c.ogeo, torch.zeros(2, c.geo.n_blocks, 3, dtype=torch.float64), ts, cp_of(leaves), 100)
    # (solve_fixed_differentiable derives the inertia from cnv: take the gradient w.r.t. a leaf the inertia does not depend on)
    val_x = sum(0.5 * float(mm) * (hist[:, 1, d] ** 2).sum() for d, mm in zip(tdofs, m_t))
    exact = torch.autograd.grad(val_x, [leaves[NAMES.index("refv")], leaves[NAMES.index("amplitude")], leaves[NAMES.index("ks")]])

    val_t, bars_t, st_f, st_r = continuous(1e-10, 1e-10)
    assert abs(val_t - float(val_x.detach())) < 1e-8 * abs(val_t)
    err_tight = max(relerr(bars_t["refv"], exact[0].numpy()), abs(bars_t["amplitude"] - exact[1].item()) / abs(exact[1].item()),
                    abs(bars_t["ks"] - exact[2].item()) / abs(exact[2].item()))
    assert err_tight < 1e-6, err_tight
    # the reference's own tolerances (problems/quads_focusing.py:73-74): the distance is whatever the controller leaves
    val_p, bars_p, st_fp, st_rp = continuous(1e-8, 1e-4)
    err_paper = max(relerr(bars_p["refv"], exact[0].numpy()), abs(bars_p["amplitude"] - exact[1].item()) / abs(exact[1].item()))
    assert err_tight < err_paper < 0.5, (err_tight, err_paper)
    assert st_rp["accepted"] < st_r["accepted"]
    print(f"continuous adjoint vs exact: tight tolerances {err_tight:.1e} ({st_f['accepted']} + {st_r['accepted']} steps), "
          f"rtol 1e-8 / atol 1e-4 {err_paper:.1e} ({st_fp['accepted']} + {st_rp['accepted']} steps)")


def test_time_cotangent_of_a_scalar_problem():
    """dy/dt = -a y: y(t) = y0 exp(-a t); L = y(t1)  =>  dL/dy0 = exp(-a t1), dL/da = -t1 y(t1), dL/dt1 = -a y(t1), dL/dt0 = +a y(t1)."""
    a, y0, t1 = 1.7, 0.8, 0.9
    func = lambda y, t: -a * y                                                    # noqa: E731
    vjp = lambda y, t, yb: (-a * yb, 0.0, np.array([float(-(y * yb).sum())]))     # noqa: E731
    ts = np.array([0.0, t1])
    ys = ref_ode.odeint(func, np.array([y0]), ts, rtol=1e-11, atol=1e-11)
    y_bar, ts_bar, a_bar = RA.odeint_rev(func, vjp, ys, ts, np.array([[0.0], [1.0]]), 1, rtol=1e-11, atol=1e-11)
    y1 = y0 * np.exp(-a * t1)
    assert abs(y_bar[0] - np.exp(-a * t1)) < 1e-9 and abs(a_bar[0] + t1 * y1) < 1e-9
    assert abs(ts_bar[1] + a * y1) < 1e-9 and abs(ts_bar[0] - a * y1) < 1e-9


def test_engine_hook_provider_equals_torch_provider(cpu_lib):
    """tests/adjoint_semantics.py feeds the same reverse pass with dfx_rhs / dfx_rhs_vjp (paper-size lattices are out of reach for
    the torch RHS): both providers must give the same right-hand side and the same vector-Jacobian products, leaf by leaf."""
    from . import adjoint_semantics as AS
    c = Case("quads", 3, True, True, seed=3, cutoff_deg=42.0, lib=cpu_lib)
    c.cp = c.cp._replace(constraint_params=FAST)
    solver = c.oracle_solver(integrator="adaptive")
    leaves = _leaves(c, solver)
    cp_of = lambda L: c.oracle_cp({k: v for k, v in zip(NAMES, L) if k != "inertia"})      # noqa: E731
    tp = RA.TorchRHS(solver, cp_of, lambda L: L[-1], leaves)
    ep = AS.EngineRHS(c.solver, c.cp)
    assert ep.n_free == tp.n_free
    rng = np.random.default_rng(0)
    y = rng.normal(size=2 * tp.n_free) * np.repeat([0.05, 20.0], tp.n_free)
    yb = rng.normal(size=2 * tp.n_free)
    t = 1.3e-4
    assert relerr(ep.func(y, t), tp.func(y, t)) < 1e-12
    vy_e, vt_e, va_e = ep.vjp(y, t, yb)
    vy_t, vt_t, va_t = tp.vjp(y, t, yb)
    assert relerr(vy_e, vy_t) < 1e-11 and abs(vt_e - vt_t) < 1e-5 * abs(vt_t)
    e, tt = ep.split(va_e), dict(zip(NAMES, tp.split(va_t)))
    assert relerr(e["centroid_node_vectors"], tt["cnv"].reshape(-1)) < 1e-10
    assert relerr(e["k"], np.array([tt["ks"], tt["ksh"], tt["kr"]]).reshape(-1)) < 1e-10
    assert relerr(e["reference_vector"], tt["refv"].reshape(-1)) < 1e-10
    assert relerr(e["damping"], tt["damping"].reshape(-1)) < 1e-10
    assert relerr(e["contact"], np.array([tt["min_angle"], tt["cutoff_angle"], tt["k_contact"]]).reshape(-1)) < 1e-9
    assert relerr(e["constraint"], np.array([tt["amplitude"], tt["input_delay"], tt["loading_rate"]]).reshape(-1)) < 1e-9
    assert relerr(e["inertia"], tt["inertia"].reshape(-1)) < 1e-10
    # block centroids cancel exactly in the angle-based contact (rounding residue in autograd), the density is not read inside rhs
    assert np.all(e["block_centroids"] == 0) and np.abs(tt["cen"]).max() < 1e-10 * np.abs(tt["cnv"]).max() and np.all(tt["density"] == 0)
