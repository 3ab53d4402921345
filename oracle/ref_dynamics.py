"""Oracle restatement of ``difflexmm/dynamics.py``, ``loading.py`` and the constrained kinematics
of ``kinematics.py`` (test infrastructure).  torch-fp64 + ``torch.autograd`` where the
reference uses ``jax.grad`` / ``jax.jacobian``; integration by ``ref_ode``.
"""
import numpy as np
import torch

from . import ref_ode
from .ref_energy import ControlParams
from .ref_geometry import F64, DOFsInfo, _t, compute_inertia


def build_constrained_kinematics(geometry, constrained_block_DOF_pairs, constrained_DOFs_fn=lambda t, **kw: 0.0):
    """kinematics.py:40-81: u = 0; u[constrained] = c(t; params); u[free] = q."""
    free_ids, con_ids, all_ids = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)
    free_t = torch.as_tensor(free_ids, dtype=torch.long)
    con_t = torch.as_tensor(con_ids, dtype=torch.long)

    def constrained_kinematics(free_DOFs, t, constraint_params={}):
        all_DOFs = torch.zeros(len(all_ids), dtype=F64)
        if len(con_ids) != 0:
            c = _t(constrained_DOFs_fn(t, **constraint_params))
            all_DOFs = all_DOFs.index_put((con_t,), c.expand(len(con_ids)) if c.ndim == 0 else c)
        all_DOFs = all_DOFs.index_put((free_t,), free_DOFs)
        return all_DOFs.reshape(geometry.n_blocks, 3)

    return constrained_kinematics


def constrain_energy(energy_fn, constrained_kinematics):
    """energy.py:473-491."""
    def constrained_energy_fn(free_DOFs, t, control_params):
        return energy_fn(constrained_kinematics(free_DOFs, t, control_params.constraint_params), control_params)
    return constrained_energy_fn


def build_loading(geometry, loaded_block_DOF_pairs, loading_fn, constrained_block_DOF_pairs=()):
    """loading.py:12-47."""
    pairs = np.asarray(loaded_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
    loaded = torch.as_tensor(pairs[:, 0] * 3 + pairs[:, 1], dtype=torch.long)
    free_ids, _, all_ids = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)
    free_t = torch.as_tensor(free_ids, dtype=torch.long)

    def global_loading_fn(state, t, loading_params):
        vec = torch.zeros(len(all_ids), dtype=F64)
        val = _t(loading_fn(state, t, **loading_params))
        vec = vec.index_put((loaded,), val.expand(len(loaded)) if val.ndim == 0 else val)
        return vec[free_t]

    return global_loading_fn


def build_viscous_damping(geometry, damped_blocks, constrained_block_DOF_pairs=()):
    """loading.py:71-106: F = -c * v with c scalar or (n_damped, 3)."""
    damped_blocks = np.asarray(damped_blocks, dtype=np.int64)
    damped = torch.as_tensor((damped_blocks[:, None] * 3 + np.arange(3)[None]).reshape(-1), dtype=torch.long)
    free_ids, _, all_ids = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)
    free_t = torch.as_tensor(free_ids, dtype=torch.long)
    ones = torch.ones((len(damped_blocks), 3), dtype=F64)

    def loading_fn(state, t, damping):
        _, velocity = state
        vec = torch.zeros(len(all_ids), dtype=F64)
        vec = vec.index_put((damped,), (_t(damping) * ones).reshape(-1))
        return -vec[free_t] * velocity

    return loading_fn


def build_RHS(energy_fn, loading_fn):
    """dynamics.py:20-57: d/dt [q, v] = [v, (-dE/dq + loading) / inertia].
    ``create_graph`` keeps the force differentiable (needed when the oracle differentiates a
    whole trajectory, where the reference would nest jax.grad)."""
    def rhs(state, t, control_params, inertia, create_graph=False):
        displacement, velocity = state[0], state[1]
        with torch.enable_grad():
            q = displacement if displacement.requires_grad else displacement.detach().requires_grad_(True)
            E = energy_fn(q, t, control_params)
            (dE,) = torch.autograd.grad(E, q, create_graph=create_graph)
        load = loading_fn(state, t, control_params.loading_params, control_params.mechanical_params.damping)
        return torch.stack([velocity, (-dE + load) / inertia])
    return rhs


def setup_dynamic_solver(geometry, energy_fn, loaded_block_DOF_pairs=None, loading_fn=None,
                         constrained_block_DOF_pairs=(), constrained_DOFs_fn=lambda t, **kw: 0.0,
                         damped_blocks=None, rtol=1e-8, atol=1e-8,
                         integrator="adaptive", steps_per_interval=None, tableau="dopri5", step_times=None):
    """dynamics.py:60-186.  ``integrator='adaptive'`` is the reference behaviour (odeint);
    ``'fixed'`` runs the same tableau on the engine's fixed grid.
    The returned solver has the reference signature and result layout (T, 2, n_blocks, 3);
    it additionally exposes ``.rhs``, ``.kinematics``, ``.free_DOF_ids``, ``.stats``."""
    kinematics = build_constrained_kinematics(geometry, constrained_block_DOF_pairs, constrained_DOFs_fn)
    constrained_energy = constrain_energy(energy_fn, kinematics)
    if loaded_block_DOF_pairs is not None and loading_fn is not None:
        _loading = build_loading(geometry, loaded_block_DOF_pairs, loading_fn, constrained_block_DOF_pairs)
    else:
        def _loading(state, t, loading_params):
            return 0.0
    if damped_blocks is not None:
        damping_fn = build_viscous_damping(geometry, damped_blocks, constrained_block_DOF_pairs)
    else:
        def damping_fn(state, t, damping):
            return 0.0

    def loading_total(state, t, loading_params, damping):
        return _loading(state, t, loading_params) + damping_fn(state, t, damping)

    rhs = build_RHS(constrained_energy, loading_total)
    free_ids, con_ids, _ = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)
    free_t = torch.as_tensor(free_ids, dtype=torch.long)
    con_t = torch.as_tensor(con_ids, dtype=torch.long)
    stats = {}

    def reduced_inertia(control_params):
        mp = control_params.mechanical_params
        if mp.inertia is None:  # dynamics.py:157-163
            full = compute_inertia(control_params.geometrical_params.centroid_node_vectors, mp.density)
        else:
            full = _t(mp.inertia)
        return full.reshape(-1)[free_t]

    def constrained_rate(t, constraint_params):
        """d c / d t, i.e. the du_dt term of dynamics.py:132-134."""
        if len(con_ids) == 0:
            return torch.zeros(0, dtype=F64)
        tt = torch.tensor(float(t), dtype=F64, requires_grad=True)
        c = _t(constrained_DOFs_fn(tt, **constraint_params))
        c = c.expand(len(con_ids)) if c.ndim == 0 else c
        if not c.requires_grad:
            return torch.zeros(len(con_ids), dtype=F64)
        rows = [torch.autograd.grad(ci, tt, retain_graph=True, allow_unused=True)[0] for ci in c]
        return torch.stack([r if r is not None else torch.zeros((), dtype=F64) for r in rows])

    def solve_dynamics(state0, timepoints, control_params):
        state0 = _t(state0)
        ts = np.asarray(timepoints, dtype=np.float64)
        y0 = state0.reshape(2, -1)[:, free_t]
        inertia = reduced_inertia(control_params)
        n_free = len(free_ids)

        def f(y, t):
            s = torch.as_tensor(y).reshape(2, n_free)
            return rhs(s, float(t), control_params, inertia).detach().numpy().reshape(-1)

        if integrator == "adaptive":
            sol = ref_ode.odeint(f, y0.detach().numpy().reshape(-1), ts, rtol=rtol, atol=atol, stats=stats)
        else:
            sol = ref_ode.odeint_fixed(f, y0.detach().numpy().reshape(-1), ts, steps_per_interval, tableau, stats=stats,
                                       step_times=step_times)
        sol = torch.as_tensor(sol).reshape(len(ts), 2, n_free)
        # dynamics.py:169-182: scatter free DOFs, constrained DOFs follow c(t) and dc/dt
        out = torch.zeros(len(ts), 2, geometry.n_blocks * 3, dtype=F64)
        for i, t in enumerate(ts):
            out[i, 0] = kinematics(sol[i, 0], float(t), control_params.constraint_params).reshape(-1).detach()
            out[i, 1, free_t] = sol[i, 1]
            if len(con_ids):
                out[i, 1, con_t] = constrained_rate(t, control_params.constraint_params).detach()
        return out.reshape(len(ts), 2, geometry.n_blocks, 3)

    solve_dynamics.rhs = rhs
    solve_dynamics.kinematics = kinematics
    solve_dynamics.free_DOF_ids = free_ids
    solve_dynamics.constrained_DOF_ids = con_ids
    solve_dynamics.reduced_inertia = reduced_inertia
    solve_dynamics.stats = stats
    return solve_dynamics


def solve_fixed_differentiable(solver, geometry, state0, timepoints, control_params, steps_per_interval,
                               tableau="dopri5", step_times=None):
    """Fixed-step solve kept on the autograd tape (small lattices only): the oracle's stand-in for
    ``jax.grad`` through ``solve_dynamics`` (problems/quads_focusing.py:565).  Returns the free-DOF
    history (T, 2, n_free) as a differentiable tensor plus the reduced inertia."""
    free_t = torch.as_tensor(solver.free_DOF_ids, dtype=torch.long)
    ts = np.asarray(timepoints, dtype=np.float64)
    y = _t(state0).reshape(2, -1)[:, free_t]
    inertia = solver.reduced_inertia(control_params)
    A = torch.as_tensor(ref_ode.BETA)
    out = [y]

    def f(s, t):
        return solver.rhs(s, t, control_params, inertia, create_graph=True)

    k1 = f(y, float(ts[0])) if tableau == "dopri5" else None
    spis = np.broadcast_to(np.asarray(steps_per_interval, dtype=np.int64), (max(len(ts) - 1, 0),))
    n = 0
    for a, b, spi in zip(ts[:-1], ts[1:], spis):
        for s in range(int(spi)):
            if step_times is None:
                h = (b - a) / int(spi)
                t = a + s * h
            else:
                t, h = float(step_times[n]), float(step_times[n + 1] - step_times[n])
            n += 1
            if tableau == "dopri5":
                ks = [k1]
                for i in range(1, 7):
                    yi = y + h * sum(A[i - 1, j] * ks[j] for j in range(i))
                    ks.append(f(yi, t + h * ref_ode.ALPHA[i - 1]))
                y = y + h * sum(ref_ode.C_SOL[j] * ks[j] for j in range(7))
                k1 = ks[6]
            else:
                k1_ = f(y, t)
                k2 = f(y + 0.5 * h * k1_, t + 0.5 * h)
                k3 = f(y + 0.5 * h * k2, t + 0.5 * h)
                k4 = f(y + h * k3, t + h)
                y = y + h / 6.0 * (k1_ + 2 * k2 + 2 * k3 + k4)
        out.append(y)
    return torch.stack(out), inertia


def solve_adaptive_replay_differentiable(solver, geometry, state0, timepoints, control_params, step_times):
    """The adaptive solve replayed on the autograd tape with its accepted step boundaries ``step_times`` (t_0 .. t_N, t_N >= the last
    output time) frozen: Dormand-Prince steps that are NOT clipped to the output times, outputs from jax's quartic dense output
    (``interp_fit_dopri`` + ``polyval``, jax.experimental.ode at the pinned 0.4.8) -- exactly the arithmetic of ``ref_ode.odeint`` for the
    decisions its controller took.  ``torch.autograd`` through the result is the exact derivative of the returned outputs with the step
    sizes held fixed: what the engine's dense-output discrete adjoint must reproduce (it differentiates neither the accept / reject
    decisions nor the step sizes; nor does the reference's continuous adjoint).  Returns (T, 2, n_free) and the reduced inertia."""
    free_t = torch.as_tensor(solver.free_DOF_ids, dtype=torch.long)
    ts = np.asarray(timepoints, dtype=np.float64)
    st = np.asarray(step_times, dtype=np.float64)
    y = _t(state0).reshape(2, -1)[:, free_t]
    inertia = solver.reduced_inertia(control_params)
    A, CS, CM = ref_ode.BETA, ref_ode.C_SOL, ref_ode.C_MID

    def f(s, t):
        return solver.rhs(s, t, control_params, inertia, create_graph=True)

    out = [y]
    k1 = f(y, float(st[0]))
    i_out = 1
    for n in range(len(st) - 1):
        t, h = float(st[n]), float(st[n + 1] - st[n])
        ks = [k1]
        for i in range(1, 7):
            yi = y + h * sum(A[i - 1, j] * ks[j] for j in range(i))
            ks.append(f(yi, t + h * ref_ode.ALPHA[i - 1]))
        y1 = y + h * sum(CS[j] * ks[j] for j in range(7))
        if i_out < len(ts) and ts[i_out] <= st[n + 1]:
            ymid = y + h * sum(CM[j] * ks[j] for j in range(7))
            dy0, dy1 = ks[0], ks[6]
            a = -2.0 * h * dy0 + 2.0 * h * dy1 - 8.0 * y - 8.0 * y1 + 16.0 * ymid
            b = 5.0 * h * dy0 - 3.0 * h * dy1 + 18.0 * y + 14.0 * y1 - 32.0 * ymid
            c = -4.0 * h * dy0 + h * dy1 - 11.0 * y - 5.0 * y1 + 16.0 * ymid
            while i_out < len(ts) and ts[i_out] <= st[n + 1]:
                r = (ts[i_out] - st[n]) / (st[n + 1] - st[n])
                out.append((((a * r + b) * r + c) * r + h * dy0) * r + y)
                i_out += 1
        y, k1 = y1, ks[6]
    if i_out != len(ts):
        raise ValueError("step_times do not reach the last output time")
    return torch.stack(out), inertia


def linear_mode_analysis(displacement, geometry, energy_fn, control_params, constrained_block_DOF_pairs=()):
    """dynamics.py:189-245: eigenvalues / eigenmodes of K q = w^2 M q around ``displacement``; K = Hessian of the constrained energy
    w.r.t. the free DOFs (autograd, where the reference calls ``jax.hessian``), generalised problem by ``scipy.linalg.eigh`` as in the
    reference, eigenvectors scaled to unit norm and scattered row-wise into (n_free, n_blocks, 3).  Also returns K."""
    import scipy.linalg
    kinematics = build_constrained_kinematics(geometry, constrained_block_DOF_pairs)
    constrained_energy = constrain_energy(energy_fn, kinematics)
    free_ids, _, all_ids = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)
    free_t = torch.as_tensor(free_ids, dtype=torch.long)
    u = _t(displacement).reshape(-1)[free_t]
    mp = control_params.mechanical_params
    if mp.inertia is None:
        inertia = compute_inertia(control_params.geometrical_params.centroid_node_vectors, mp.density).reshape(-1)[free_t]
    else:
        inertia = _t(mp.inertia).reshape(-1)[free_t]
    K = torch.autograd.functional.hessian(lambda q: constrained_energy(q, 0.0, control_params), u).detach().numpy()
    w2, vec = scipy.linalg.eigh(K, np.diag(inertia.detach().numpy()))
    vec = (vec / np.linalg.norm(vec, axis=0)).T
    modes = np.zeros((len(free_ids), len(all_ids)))
    modes[:, np.asarray(free_ids)] = vec
    return w2, modes.reshape(len(free_ids), geometry.n_blocks, 3), K
