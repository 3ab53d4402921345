"""Method of moving asymptotes for the design loop.

The reference maximises its objectives with NLopt's ``LD_MMA`` (``problems/quads_focusing.py:546-652``:
``nlopt.opt(nlopt.LD_MMA, n)``, ``add_inequality_mconstraint``, ``set_max_objective``, bounds, ``maxeval``).  NLopt is a
third-party dependency (2.7.1, ``poetry.lock``) that is not on the target image; this is a NumPy restatement of the
algorithm it documents: Svanberg's globally convergent MMA / CCSA (K. Svanberg, SIAM J. Optim. 12(2), 2002) in the form
NLopt uses it --

  * every function (objective and each constraint) is replaced around the current point x by the separable convex
    approximation  g(x + d) = f(x) + sum_j [ f'_j s_j^2 d_j + (|f'_j| s_j + rho / 2) d_j^2 ] / (s_j^2 - d_j^2),
    with moving asymptote widths s_j and a conservativeness parameter rho per function (nlopt/mma.c dual_func:
    v = |df| sigma + rho / 2; the rho update uses w = sum_j d_j^2 / (s_j^2 - d_j^2) / 2, so that one update of rho makes the
    approximation exactly conservative at the rejected candidate);
  * the approximate problem is solved through its dual (one multiplier per constraint; the inner minimisation over d is
    separable and closed-form), here with bound-constrained L-BFGS on the concave dual, restricted to a working set of
    constraints that is grown until nothing outside it is violated (the lattices have thousands of slack constraints);
  * the candidate is accepted when every approximation is conservative at it (g >= f), otherwise the rho of the
    offending functions grow and the sub-problem is solved again (inner iterations);
  * after an accepted step rho shrinks and s_j shrinks for oscillating coordinates / grows for monotone ones.

Constraint Jacobians may be ``scipy.sparse`` matrices (the geometric constraints of the lattices are local).
Host code (a few thousand variables, once per design iteration), not part of the kernel path.
"""
import numpy as np
import scipy.optimize
import scipy.sparse as sp


# initial working set of the dual: constraints whose value is within this fraction of the largest |constraint| of being active (the
# set is grown until the sub-problem's solution violates nothing outside it, so the margin only trades passes for dimension:
# L-BFGS-B's own cost grows with the dimension -- 5 % put ~2 000 of the 4 448 geometric constraints of the 24x16 lattice into every
# dual solve, 0.56 s of host time per member and 4 evaluations; 0.5 %: see profiles/r02_mma_working_set.txt)
WORKING_SET_MARGIN = 0.005
WORKING_SET_GROWTH = 256


class MMAResult(dict):
    def __getattr__(self, name):
        if name.startswith("__"):            # pickle / copy probe for optional protocol methods
            raise AttributeError(name)
        return self.get(name)


def _abs(J):
    return abs(J) if sp.issparse(J) else np.abs(J)


def mma_steps(x0, lower=None, upper=None, constraints=(), maxeval=100, ftol_rel=0.0, xtol_rel=0.0,
              callback=None, verbose=False, constraint_tol=0.0):
    """The algorithm as a coroutine: ``yield x`` asks the driver for ``(value, gradient)`` of the objective at x
    (``generator.send((f, g))``); the generator returns the :class:`MMAResult`.  Lets many optimisations advance in
    lock-step with their objective evaluations batched into one engine call (:func:`mma_minimize_ensemble`).
    ``constraint_tol``: a point counts as feasible when every constraint is <= this (the ``tol`` argument of NLopt's
    ``add_inequality_mconstraint``; it only enters the feasibility judgement, the sub-problems use the constraints as given)."""
    x = np.array(x0, dtype=float).ravel()
    n = x.size
    lb = np.full(n, -np.inf) if lower is None else np.broadcast_to(np.asarray(lower, dtype=float), (n,)).copy()
    ub = np.full(n, np.inf) if upper is None else np.broadcast_to(np.asarray(upper, dtype=float), (n,)).copy()
    x = np.clip(x, lb, ub)
    finite = np.isfinite(lb) & np.isfinite(ub)
    sigma = np.where(finite, 0.5 * (ub - lb), 1.0)
    sig_min = np.where(finite, 1e-8 * (ub - lb), 1e-8)
    sig_max = np.where(finite, 10.0 * (ub - lb), np.inf)

    def eval_constraints(z):
        if not constraints:
            return np.zeros(0), sp.csr_matrix((0, n))
        vals, jacs = [], []
        for c, jac in constraints:
            vals.append(np.asarray(c(z), dtype=float).ravel())
            jacs.append(sp.csr_matrix(jac(z)))
        return np.concatenate(vals), sp.vstack(jacs).tocsr()

    def eval_constraint_values(z):
        return np.concatenate([np.asarray(c(z), dtype=float).ravel() for c, _ in constraints]) if constraints else np.zeros(0)

    f, g = yield x
    g = np.asarray(g, dtype=float).ravel()
    n_eval = 1
    fc, J = eval_constraints(x)
    m = fc.size
    rho, rhoc = 1.0, np.ones(m)
    feasible = bool(np.all(fc <= constraint_tol))
    infeasibility = fc.max() if m else 0.0
    x_prev = x_prev2 = x.copy()
    best = (f, x.copy()) if feasible else (np.inf, x.copy())
    y = np.zeros(m)
    history = [f]
    k = 0
    status = "maxeval"
    while n_eval < maxeval:
        k += 1
        s2 = sigma * sigma
        Jabs = _abs(J)
        lo = np.maximum(lb - x, -0.9 * sigma)
        hi = np.minimum(ub - x, 0.9 * sigma)
        accepted = False
        for inner in range(60):
            # ---- dual of the separable approximation
            def primal(yv):
                u = s2 * (g + (J.T @ yv if m else 0.0))
                v = sigma * (np.abs(g) + (Jabs.T @ yv if m else 0.0)) + 0.5 * (rho + (rhoc @ yv if m else 0.0))
                with np.errstate(divide="ignore", invalid="ignore"):
                    d = np.where(np.abs(u) < 1e-3 * (v + 1e-300), -0.5 * u / np.maximum(v, 1e-300) * 1.0,
                                 (s2 / u) * (-v + np.sqrt(np.maximum(v * v - u * u / s2, 0.0))))
                d = np.where(np.isfinite(d), d, 0.0)
                return np.clip(d, lo, hi)

            def approx(d, grad, r):
                """g(x+d) - f(x) for one function with gradient grad and conservativeness r."""
                return np.sum((grad * s2 * d + (np.abs(grad) * sigma + 0.5 * r) * d * d) / (s2 - d * d))

            def approx_constraints(d):
                w = d / (s2 - d * d)
                w2 = d * d / (s2 - d * d)
                return fc + J @ (s2 * w) + Jabs @ (sigma * w2) + 0.5 * rhoc * np.sum(w2)

            if m:
                # The geometric constraints number in the thousands and almost all of them are slack: solve the dual on a
                # working set (multipliers of the others are zero) and grow it until the approximate problem's solution
                # violates no constraint outside it -- the same KKT point as the dual over all of them.
                active = (y > 0.0) | (fc > -WORKING_SET_MARGIN * max(1e-300, np.abs(fc).max()))
                for _pass in range(20):
                    idx = np.flatnonzero(active)
                    if idx.size:
                        # the dual is evaluated ~100 times per solve: rows of the working set sliced and transposed ONCE (CSR both
                        # ways) -- the full (m, n) products and the transposed-matrix objects SciPy builds per `J.T @ y` call
                        # were two thirds of the host time of the config-5 ensemble (profiles/r02_c5_host_profile.txt)
                        Ji, Jai = J[idx].tocsr(), Jabs[idx].tocsr()
                        JiT, JaiT = Ji.T.tocsr(), Jai.T.tocsr()
                        fci, rci, gabs = fc[idx], rhoc[idx], np.abs(g)

                        def neg_dual(ya):
                            u = s2 * (g + JiT @ ya)
                            v = sigma * (gabs + JaiT @ ya) + 0.5 * (rho + rci @ ya)
                            with np.errstate(divide="ignore", invalid="ignore"):
                                d = np.where(np.abs(u) < 1e-3 * (v + 1e-300), -0.5 * u / np.maximum(v, 1e-300),
                                             (s2 / u) * (-v + np.sqrt(np.maximum(v * v - u * u / s2, 0.0))))
                            d = np.clip(np.where(np.isfinite(d), d, 0.0), lo, hi)
                            den = s2 - d * d
                            w, w2 = d / den, d * d / den
                            gc = fci + Ji @ (s2 * w) + Jai @ (sigma * w2) + 0.5 * rci * np.sum(w2)
                            val = f + np.sum((g * s2 * d + (gabs * sigma + 0.5 * rho) * d * d) / den) + ya @ gc
                            return -val, -gc
                        res = scipy.optimize.minimize(neg_dual, y[idx], jac=True, method="L-BFGS-B",
                                                      bounds=scipy.optimize.Bounds(np.zeros(idx.size), np.full(idx.size, np.inf)),
                                                      options=dict(maxiter=100, ftol=1e-12, gtol=1e-8))
                        y = np.zeros(m); y[idx] = np.maximum(res.x, 0.0)
                    else:
                        y = np.zeros(m)
                    excess = np.where(active, -np.inf, approx_constraints(primal(y)))
                    n_violated = int(np.count_nonzero(excess > 1e-10))
                    if not n_violated:
                        break
                    # the worst offenders first, at most doubling the set per pass: the unconstrained step of a wide-asymptote
                    # iteration violates nearly everything, yet few of those constraints carry a multiplier at the solution
                    n_add = min(n_violated, max(WORKING_SET_GROWTH, idx.size)) if _pass < 18 else n_violated
                    active[np.argpartition(excess, -n_add)[-n_add:]] = True
            d = primal(y)
            x_new = x + d
            f_new, g_new = yield x_new
            n_eval += 1
            fc_new = eval_constraint_values(x_new)
            g0 = f + approx(d, g, rho)
            gc = approx_constraints(d) if m else np.zeros(0)
            feas_new = bool(np.all(fc_new <= constraint_tol))
            infeas_new = fc_new.max() if m else 0.0
            # NLopt's acceptance of a new best point: feasible and better, or less infeasible while none is feasible yet
            if (feas_new and (f_new < best[0] or not feasible)) or (not feasible and infeas_new < infeasibility):
                best = (f_new if feas_new else np.inf, x_new.copy())
                feasible = feasible or feas_new
                infeasibility = min(infeasibility, infeas_new)
            conservative = g0 >= f_new - 1e-12 * max(1.0, abs(f_new)) and (not m or np.all(gc >= fc_new - 1e-12))
            if conservative or n_eval >= maxeval:
                accepted = True
                break
            w = 0.5 * np.sum(d * d / (s2 - d * d))
            if w <= 0:
                accepted = True
                break
            if g0 < f_new:
                rho = min(10.0 * rho, 1.1 * (rho + (f_new - g0) / w))
            if m:
                bad = gc < fc_new
                rhoc[bad] = np.minimum(10.0 * rhoc[bad], 1.1 * (rhoc[bad] + (fc_new[bad] - gc[bad]) / w))
        if not accepted:
            status = "inner iterations exhausted"
            break
        # ---- outer update
        x_prev2, x_prev = x_prev, x
        x, f_old, f, g = x_new, f, f_new, np.asarray(g_new, dtype=float).ravel()
        fc, J = eval_constraints(x)
        history.append(f)
        if callback is not None:
            callback(x, f, fc)
        if verbose:
            print(f"mma {k}: f = {f:.6e}  max constraint = {fc.max() if m else 0.0:.3e}  evals = {n_eval}")
        rho = max(0.1 * rho, 1e-5)
        rhoc = np.maximum(0.1 * rhoc, 1e-5)
        if k > 1:
            osc = (x - x_prev) * (x_prev - x_prev2)
            sigma = np.clip(sigma * np.where(osc < 0, 0.7, np.where(osc > 0, 1.2, 1.0)), sig_min, sig_max)
        if ftol_rel > 0 and abs(f - f_old) <= ftol_rel * max(abs(f), abs(f_old)):
            status = "ftol"
            break
        if xtol_rel > 0 and np.all(np.abs(x - x_prev) <= xtol_rel * np.maximum(np.abs(x), 1e-300)):
            status = "xtol"
            break
    x_best = best[1] if np.isfinite(best[0]) else x
    f_best = best[0] if np.isfinite(best[0]) else f
    return MMAResult(x=x_best, fun=f_best, n_eval=n_eval, n_iter=k, status=status, feasible=feasible, history=history)


def _drive(gen, fun):
    try:
        x = next(gen)
        while True:
            x = gen.send(fun(x))
    except StopIteration as stop:
        return stop.value


def mma_minimize(fun, x0, **kw):
    """minimise fun(x) -> (value, gradient) subject to lower <= x <= upper and c(x) <= 0 for every
    ``(c, jac)`` pair in ``constraints`` (c(x) -> (m,), jac(x) -> (m, n) dense or sparse).
    ``maxeval`` counts objective evaluations like NLopt's ``set_maxeval``."""
    return _drive(mma_steps(x0, **kw), fun)


def maximizing(gen):
    """A minimising coroutine turned into a maximising one: values and gradients are negated on the way in, the result on the
    way out (``opt.set_max_objective``)."""
    x = next(gen)
    try:
        while True:
            f, g = yield x
            x = gen.send((-f, -np.asarray(g)))
    except StopIteration as stop:
        res = stop.value
        res["fun"] = -res["fun"]
        res["history"] = [-h for h in res["history"]]
        return res


def _worker_main(conn):
    """One host process of :class:`MemberWorkers`: owns the coroutines of some members and advances them on request."""
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    gens = {}

    def advance(idx, value):
        try:
            return idx, False, (next(gens[idx]) if value is None else gens[idx].send(value))
        except StopIteration as stop:
            del gens[idx]
            return idx, True, stop.value

    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        if msg[0] == "close":
            return
        try:
            if msg[0] == "new":
                out = []
                for idx, factory, args, kwargs in msg[1]:
                    gens[idx] = factory(*args, **kwargs)
                    out.append(advance(idx, None))
            else:                                   # "send": [(idx, (value, gradient)), ...]
                out = [advance(idx, value) for idx, value in msg[1]]
            conn.send(("ok", out))
        except Exception as e:                      # the parent re-raises: a failing member must not hang the ensemble
            import traceback
            conn.send(("error", f"{e!r}\n{traceback.format_exc()}"))


class MemberWorkers:
    """Host processes for the per-member work of a lock-step ensemble (constraint evaluations, MMA sub-problems: a few
    milliseconds of NumPy / L-BFGS-B per member and round, 15 of the 28 s of a 256-member x 4-evaluation run of config 5 when
    done in the engine's process: profiles/r02_c5_host_profile_v2.txt).  Members are dealt to the workers once; per round each
    worker receives the (value, gradient) of its members and returns their next points.

    Create the workers BEFORE the process touches the GPU (``fork`` start method: a forked child of a process with a live HIP
    runtime must never call into it, and nothing here does, but forking first keeps the children free of its state)."""

    def __init__(self, n_workers):
        import multiprocessing as mp
        import queue
        import threading
        ctx = mp.get_context("fork")
        self._conns, self._procs, self._replies, self._readers = [], [], [], []
        for _ in range(max(1, int(n_workers))):
            parent, child = ctx.Pipe()
            proc = ctx.Process(target=_worker_main, args=(child,), daemon=True)
            proc.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(proc)
        # Replies are drained by one reader thread per worker (started after every fork: the children stay single-threaded).  Without
        # them the pipelined ensemble deadlocks on large messages: post() blocks in Connection.send to a worker that is itself blocked
        # sending the reply of the OTHER half, which the caller has not collected yet (a 128 x 128 design is 528 KB, a pipe buffer 208 KB).
        for conn in self._conns:
            q = queue.Queue()
            t = threading.Thread(target=self._drain, args=(conn, q), daemon=True)
            t.start()
            self._replies.append(q)
            self._readers.append(t)
        self._owner = {}

    @staticmethod
    def _drain(conn, q):
        while True:
            try:
                q.put(conn.recv())
            except (EOFError, OSError, ValueError, TypeError):     # TypeError: the handle of a connection closed under the reader is None
                q.put(("error", "the worker's connection closed"))
                return

    def __len__(self):
        return len(self._procs)

    def _collect(self, used):
        out = {}
        for w in used:
            status, payload = self._replies[w].get()
            if status != "ok":
                raise RuntimeError(f"ensemble worker {w} failed: {payload}")
            for idx, done, value in payload:
                out[idx] = (done, value)
        return out

    def start(self, specs):
        """specs[i] = (factory, args, kwargs), picklable; returns {i: (done, first point or result)}."""
        per = {}
        for idx, spec in enumerate(specs):
            w = idx % len(self._conns)
            self._owner[idx] = w
            per.setdefault(w, []).append((idx,) + tuple(spec))
        for w, items in per.items():
            self._conns[w].send(("new", items))
        return self._collect(per)

    def post(self, values):
        """Hand {i: (value, gradient)} to the workers that own those members and return at once (a ticket for :meth:`collect`):
        the workers run the members' next MMA steps while the caller does something else -- integrate the other half of the
        ensemble on the device."""
        per = {}
        for idx, v in values.items():
            per.setdefault(self._owner[idx], []).append((idx, v))
        for w, items in per.items():
            self._conns[w].send(("send", items))
        return per

    def collect(self, ticket):
        """{i: (done, next point or result)} of the members of a :meth:`post`."""
        return self._collect(ticket)

    def send(self, values):
        """values: {i: (value, gradient)} of the members still running; returns {i: (done, next point or result)}."""
        return self.collect(self.post(values))

    def close(self):
        for c in self._conns:
            try:
                c.send(("close",))
                c.close()
            except (OSError, BrokenPipeError):
                pass
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        for t in self._readers:
            t.join(timeout=1)
        self._conns, self._procs, self._replies, self._readers = [], [], [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def drive_ensemble(batch_fun, specs, workers=None, groups=1):
    """Independent optimisations advancing in lock-step.  ``specs[i] = (factory, args, kwargs)``: ``factory(*args, **kwargs)``
    is member i's coroutine (``yield x`` asks for ``(value, gradient)`` at x, its return value is the member's result).
    ``batch_fun(list of x) -> list of (value, gradient)`` is called once per round with the pending point of EVERY member
    (members that have finished resubmit their last point so the batch keeps its size: the engine integrates a fixed number of
    members side by side).  With ``workers`` (:class:`MemberWorkers`) the coroutines live in host processes; the sequence of
    iterates of every member is the same either way, and the same as if it ran alone.

    ``groups=2`` (needs ``workers``): the members are split into two halves that take turns on the device -- while one half is
    integrated (``batch_fun`` is then called with that half's points only), the workers run the MMA steps of the other half, so the
    host work of a round hides behind the device work of the next.  Every member still sees exactly its own sequence of
    evaluations: results are identical to ``groups=1``."""
    n = len(specs)
    results = [None] * n
    pending = [None] * n
    if groups > 1:
        if workers is None or groups != 2 or n % 2:
            raise ValueError("drive_ensemble: groups=2 needs host workers and an even number of members")
        return _drive_two_groups(batch_fun, specs, workers, results, pending)

    def absorb(replies):
        for i, (done, value) in replies.items():
            if done:
                results[i] = value
            else:
                pending[i] = value

    if workers is None:
        gens = [f(*a, **k) for f, a, k in specs]

        def advance(i, value):
            try:
                return False, (next(gens[i]) if value is None else gens[i].send(value))
            except StopIteration as stop:
                return True, stop.value
        absorb({i: advance(i, None) for i in range(n)})
    else:
        absorb(workers.start(specs))
    while any(r is None for r in results):
        values = batch_fun(pending)
        live = {i: values[i] for i in range(n) if results[i] is None}
        absorb({i: advance(i, v) for i, v in live.items()} if workers is None else workers.send(live))
    return results


def _drive_two_groups(batch_fun, specs, workers, results, pending):
    n = len(specs)
    halves = [list(range(0, n // 2)), list(range(n // 2, n))]

    def absorb(replies):
        for i, (done, value) in replies.items():
            if done:
                results[i] = value
            else:
                pending[i] = value

    absorb(workers.start(specs))
    ticket = [None, None]
    g = 0
    while any(r is None for r in results):
        if ticket[g] is not None:                       # this half's next points: computed while the other half was on the device
            absorb(workers.collect(ticket[g]))
            ticket[g] = None
        ids = halves[g]
        if any(results[i] is None for i in ids):
            values = batch_fun([pending[i] for i in ids], ids)
            live = {i: values[k] for k, i in enumerate(ids) if results[i] is None}
            ticket[g] = workers.post(live)
        g ^= 1
    for t in ticket:
        if t is not None:
            absorb(workers.collect(t))
    return results


def _mma_member(x0, maximize, kw):
    gen = mma_steps(x0, **kw)
    return maximizing(gen) if maximize else gen


def mma_minimize_ensemble(batch_fun, x0s, per_member_kw=None, workers=None, _maximize=False, **kw):
    """:func:`mma_minimize` for many members in lock-step (:func:`drive_ensemble`).  With ``workers`` the keyword arguments
    (constraint callables included) must be picklable."""
    specs = [(_mma_member, (x0, _maximize, dict(kw, **(per_member_kw[i] if per_member_kw else {}))), {}) for i, x0 in enumerate(x0s)]
    return drive_ensemble(batch_fun, specs, workers)


def mma_maximize_ensemble(batch_fun, x0s, **kw):
    return mma_minimize_ensemble(batch_fun, x0s, _maximize=True, **kw)


def mma_maximize(fun, x0, **kw):
    """maximise: fun(x) -> (value, gradient); same options as :func:`mma_minimize` (``opt.set_max_objective``)."""
    return _drive(maximizing(mma_steps(x0, **kw)), fun)
