"""TEST INFRASTRUCTURE -- restatement of the REVERSE pass of ``jax.experimental.ode.odeint`` (jax 0.4.8, ``_odeint_rev``): what
``jax.grad`` through the reference's ``solve_dynamics`` (``difflexmm/dynamics.py:166``, differentiated at
``problems/quads_focusing.py:565``) actually computes -- the CONTINUOUS adjoint, integrated backwards with the same adaptive
Dormand-Prince controller, interval by interval between the output times.

Third-party code, not under ``/root/reference`` (pinned ``jax 0.4.8``, ``poetry.lock:614-615``); restated from its published algorithm
(SURVEY appendix B.1):

    residuals (ys, ts, args);  y_bar = g[-1], t0_bar = 0, args_bar = 0
    for i = T-1 .. 1:
        t_bar_i = f(ys[i], ts[i]) . g[i];      t0_bar -= t_bar_i
        integrate the augmented state (y, y_bar, t0_bar, args_bar) with dynamics (-f, vjp_y(y_bar), vjp_t(y_bar), vjp_args(y_bar))
        in negative time from -ts[i] to -ts[i-1] with ``odeint`` itself (fresh initial-step selection every interval; the error
        norm runs over the WHOLE augmented vector, parameters included; dense output at -ts[i-1])
        y_bar += g[i-1]
    returns (y_bar, ts_bar = [t0_bar, t_bar_1 .. t_bar_{T-1}], args_bar)

Parity pinning: no JAX here, so this restatement is pinned only by its own consistency checks (``tests/test_oracle_adjoint.py``: at
tight tolerances it converges to the gradient of the discretised solve obtained by autograd).  It exists to MEASURE how far the
reference's own gradient sits from the exact one at the tolerances the reference's problems use, next to the engine's discrete adjoint.
"""
import numpy as np
import torch

from . import ref_ode

F64 = torch.float64


def odeint_rev(func, vjp, ys, ts, g, args_size, rtol=1.4e-8, atol=1.4e-8, mxstep=np.inf, hmax=np.inf, stats=None):
    """``func(y, t) -> dy``; ``vjp(y, t, y_bar) -> (y_bar . df/dy, y_bar . df/dt, y_bar . df/dargs)`` on flat float64 arrays
    (``args_size`` entries in the last one).  ``ys`` (T, n): the forward solution at ``ts``; ``g`` (T, n): cotangent of ``ys``.
    Returns (y0_bar, ts_bar, args_bar)."""
    ys, ts, g = np.asarray(ys, dtype=np.float64), np.asarray(ts, dtype=np.float64), np.asarray(g, dtype=np.float64)
    n = ys.shape[1]

    def aug_dynamics(aug, s):
        # `s` is negative time: negate again to get back to normal time
        y, y_bar = aug[:n], aug[n:2 * n]
        t = -s
        y_dot = func(y, t)
        vy, vt, va = vjp(y, t, y_bar)
        return np.concatenate([-y_dot, vy, [vt], va])

    y_bar, t0_bar, args_bar = g[-1].copy(), 0.0, np.zeros(args_size)
    rev_ts_bar = []
    n_try = n_acc = 0
    for i in range(len(ts) - 1, 0, -1):
        t_bar = float(np.dot(func(ys[i], ts[i]), g[i]))          # effect of moving the measurement time
        t0_bar = t0_bar - t_bar
        st = {}
        sol = ref_ode.odeint(aug_dynamics, np.concatenate([ys[i], y_bar, [t0_bar], args_bar]), np.array([-ts[i], -ts[i - 1]]),
                             rtol=rtol, atol=atol, mxstep=mxstep, hmax=hmax, stats=st)
        n_try += st["attempted"]; n_acc += st["accepted"]
        aug = sol[1]
        y_bar, t0_bar, args_bar = aug[n:2 * n] + g[i - 1], float(aug[2 * n]), aug[2 * n + 1:]
        rev_ts_bar.append(t_bar)
    if stats is not None:
        stats.update(attempted=n_try, accepted=n_acc)
    return y_bar, np.concatenate([[t0_bar], rev_ts_bar[::-1]]), args_bar


class TorchRHS:
    """``func`` / ``vjp`` for :func:`odeint_rev` from the oracle's torch RHS (``ref_dynamics.build_RHS``): the VJPs are
    ``torch.autograd`` through the force, which is itself an autograd gradient (``create_graph=True``) -- where jax nests ``jax.vjp``
    around ``jax.grad``.  ``leaves``: list of tensors (requires_grad) that ``control_params_of(leaves)`` / ``inertia_of(leaves)`` are
    built from: the flattened ``args`` of the reference's ``odeint(rhs, state0, timepoints, control_params, inertia)`` call."""

    def __init__(self, solver, control_params_of, inertia_of, leaves):
        self.rhs, self.leaves = solver.rhs, list(leaves)
        self.cp_of, self.inertia_of = control_params_of, inertia_of
        self.sizes = [int(l.numel()) for l in self.leaves]
        self.args_size = int(sum(self.sizes))
        self.n_free = len(solver.free_DOF_ids)
        self.evals = 0

    def func(self, y, t):
        self.evals += 1
        with torch.no_grad():
            cp, inertia = self.cp_of(self.leaves), self.inertia_of(self.leaves)
        s = torch.as_tensor(np.asarray(y, dtype=np.float64)).reshape(2, self.n_free)
        return self.rhs(s, float(t), cp, inertia).detach().numpy().reshape(-1)

    def vjp(self, y, t, y_bar):
        self.evals += 1
        yt = torch.tensor(np.asarray(y, dtype=np.float64).reshape(2, self.n_free), requires_grad=True)
        tt = torch.tensor(float(t), dtype=F64, requires_grad=True)
        cp, inertia = self.cp_of(self.leaves), self.inertia_of(self.leaves)
        dy = self.rhs(yt, tt, cp, inertia, create_graph=True)
        ins = [yt, tt] + self.leaves
        out = torch.autograd.grad(dy, ins, grad_outputs=torch.as_tensor(np.asarray(y_bar, dtype=np.float64)).reshape(2, self.n_free),
                                  allow_unused=True)
        vy = out[0].numpy().reshape(-1) if out[0] is not None else np.zeros(2 * self.n_free)
        vt = float(out[1]) if out[1] is not None else 0.0
        va = np.concatenate([(o.detach().numpy().reshape(-1) if o is not None else np.zeros(sz)) for o, sz in zip(out[2:], self.sizes)]) \
            if self.leaves else np.zeros(0)
        return vy, vt, va

    def split(self, args_bar):
        out, k = [], 0
        for l, sz in zip(self.leaves, self.sizes):
            out.append(np.asarray(args_bar[k:k + sz]).reshape(tuple(l.shape)))
            k += sz
        return out
