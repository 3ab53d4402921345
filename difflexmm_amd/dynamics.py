"""``setup_dynamic_solver`` with the reference's signature (``difflexmm/dynamics.py:60-186``) on top of the HIP
engine.  The returned ``solve_dynamics(state0, timepoints, control_params)`` has the reference's shapes,
``(T, 2, n_blocks, 3)``, and additionally ``solve_dynamics.vjp(fields_bar)`` -- the counterpart of taking
``jax.grad`` through the reference solver -- returning a ``ControlParams``-shaped gradient tree.

Differences that are inherent to running inside hand-written kernels (all keyword-only, all with defaults):
  * by default the forward solve is the reference's own scheme: adaptive Dormand-Prince 5(4) controlled by ``rtol`` /
    ``atol`` with dense output at ``timepoints`` (jax.experimental.ode.odeint semantics, every member its own step).
    With ``steps_per_interval=k`` (an int, or one int per output interval) the same tableau runs on k equal steps
    between consecutive ``timepoints`` (``step_times=`` gives the step boundaries explicitly).  ``keep_trajectory=True`` (needed by ``vjp``: the reverse sweep is the exact
    discrete adjoint of a FIXED grid) without a grid first runs the adaptive controller and then freezes ITS grid: the
    step boundaries it accepted (for the member that needed the most steps) plus the output times, so the
    differentiated solve has the accuracy rtol / atol ask for;
  * ``energy_fn``, ``loading_fn`` and ``constrained_DOFs_fn`` must come from ``difflexmm_amd.energy`` /
    ``difflexmm_amd.loading`` (declarative specs), otherwise ``TypeError`` at setup;
  * ``batch=B`` integrates B members (list of B ``ControlParams``) side by side.
"""
import os
from typing import Optional

import numpy as np

from . import _binding as _b
from .energy import _EnergyFn
from .geometry import (DOFsInfo, compute_inertia, compute_inertia_vjp, void_angles0, void_angles0_vjp)
from .loading import as_time_function, zero
from .utils import (ContactParams, ControlParams, GeometricalParams, MechanicalParams)


_FLAT_CACHE = {}      # id(centroid_node_vectors) -> (weakref, bonds, density, inertia, void_angle0)


def remember_flat(cnv, bonds, density, inertia, void_angle0):
    """What a solver derives from a design's node vectors, computed elsewhere (the native design map, problems.prefetch_designs): kept for
    ``DynamicSolver._flatten`` under the identity of the (read-only) array, like the entries it makes itself."""
    import weakref
    key = id(cnv)
    if len(_FLAT_CACHE) > 1024:
        _FLAT_CACHE.clear()
    _FLAT_CACHE[key] = (weakref.ref(cnv, lambda _r, k=key: _FLAT_CACHE.pop(k, None)), bonds, np.array(density, copy=True), inertia, void_angle0)


def _same_scalars(a, b):
    try:
        if isinstance(a, tuple) or isinstance(b, tuple):
            return isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b) and all(_same_scalars(x, y) for x, y in zip(a, b))
        if a is None or b is None:
            return a is b
        return np.ndim(a) == 0 and np.ndim(b) == 0 and float(a) == float(b)
    except (TypeError, ValueError):
        return False


def _bcast(x, n):
    return np.broadcast_to(np.asarray(x, dtype=float), (n,)).copy()


class DynamicSolver:
    """Callable returned by :func:`setup_dynamic_solver`."""

    def __init__(self, geometry, energy_fn, loaded_block_DOF_pairs, loading_fn, constrained_block_DOF_pairs,
                 constrained_DOFs_fn, damped_blocks, rtol, atol, integrator, steps_per_interval, batch, device, lib, streams=0, grid_refine=1):
        if not isinstance(energy_fn, _EnergyFn) or energy_fn.spec.bond_model is None:
            raise TypeError("energy_fn must be built with difflexmm_amd.energy.build_strain_energy "
                            "(optionally combined with build_contact_energy)")
        self.geometry = geometry
        self.spec = energy_fn.spec
        self.n_blocks, self.n_npb = geometry.n_blocks, geometry.n_npb
        self.bonds = self.spec.bond_connectivity
        self.rtol, self.atol = rtol, atol
        self.steps_per_interval = steps_per_interval
        if int(grid_refine) < 1:
            raise ValueError("grid_refine must be >= 1")
        self.grid_refine = int(grid_refine)
        self.max_attempts = 10_000_000          # budget of the adaptive controller (attempted steps per member): jax's mxstep is unbounded
        self.batch = int(batch)
        self.damped_blocks = None if damped_blocks is None else np.asarray(damped_blocks, dtype=np.int64)
        self.constrained_pairs = np.asarray(constrained_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
        self.free_DOF_ids, self.constrained_DOF_ids, _ = DOFsInfo(self.n_blocks, self.constrained_pairs)
        self.constraint_fn = as_time_function(constrained_DOFs_fn, "constrained_DOFs_fn")
        if loaded_block_DOF_pairs is not None and loading_fn is not None:
            self.loaded_pairs = np.asarray(loaded_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
            self.loading_fn = as_time_function(loading_fn, "loading_fn")
        else:
            self.loaded_pairs = np.zeros((0, 2), dtype=np.int64)
            self.loading_fn = zero
        # time-function slots: constraint terms first, then loading terms
        self.con_terms, self.load_terms = self.constraint_fn.terms, self.loading_fn.terms
        n_fns = len(self.con_terms) + len(self.load_terms)
        if n_fns > _b.DFX_MAX_FNS:
            raise ValueError(f"at most {_b.DFX_MAX_FNS} time functions (constraint + loading terms) are supported")
        fn_types = [f.type_id for f in self.con_terms + self.load_terms]
        # blocks with constrained or loaded DOFs
        special = {}

        def entry(block):
            return special.setdefault(int(block), [0, np.zeros((3, _b.DFX_MAX_FNS)), np.zeros((3, _b.DFX_MAX_FNS))])

        n_con = len(self.constrained_pairs)
        for j, (blk, d) in enumerate(self.constrained_pairs):
            e = entry(blk)
            e[0] |= 1 << int(d)
            for f, term in enumerate(self.con_terms):
                e[1][d, f] = _bcast(term.vector, n_con)[j]
        n_load = len(self.loaded_pairs)
        for j, (blk, d) in enumerate(self.loaded_pairs):
            e = entry(blk)
            for f, term in enumerate(self.load_terms):
                e[2][d, len(self.con_terms) + f] = _bcast(term.vector, n_load)[j]
        self._special = [(blk, e[0], e[1], e[2]) for blk, e in sorted(special.items())]
        self.engine = _b.Engine(self.n_blocks, self.n_npb, self.bonds, self.spec.bond_model,
                                int(self.spec.contact or 0), self._special, fn_types,
                                batch=self.batch, tableau=integrator, device=device, lib=lib,
                                fn_tables=[getattr(f, "table", None) for f in self.con_terms + self.load_terms], streams=streams)
        self._last = None
        self.solve_count = 0          # forward solves run so far (what the engine's resident history belongs to)

    # -- ControlParams -> engine arrays -------------------------------------------------------------
    def _memo(self, name, source, build):
        """The flattened form of a ControlParams leaf that rarely changes between members and evaluations (stiffnesses, reference vectors,
        damping): rebuilt only when the leaf is another object (arrays) or another value (scalars / tuples of scalars)."""
        cache = self.__dict__.setdefault("_memo_cache", {})
        hit = cache.get(name)
        same = hit is not None and (hit[0] is source or (not isinstance(source, np.ndarray) and not isinstance(hit[0], np.ndarray) and _same_scalars(hit[0], source)))
        if same:
            return hit[1]
        arr = build()
        arr.flags.writeable = False
        cache[name] = (source, arr)
        return arr

    def _stack(self, flats):
        """The members' flattened arrays as batch-leading arrays in buffers this solver keeps: a row is copied only when its source is
        another object than last time (the static leaves above, and designs that did not change, cost nothing)."""
        bufs = self.__dict__.setdefault("_batch_bufs", {})
        out = {}
        for k in flats[0]:
            rows = [f[k] for f in flats]
            shape = (len(rows),) + np.shape(rows[0])
            buf = bufs.get(k)
            if buf is None or buf[0].shape != shape:
                buf = bufs[k] = [np.empty(shape), [None] * len(rows)]
            for m, r in enumerate(rows):
                if buf[1][m] is not r or not isinstance(r, np.ndarray) or r.flags.writeable:
                    buf[0][m] = r
                    buf[1][m] = r if isinstance(r, np.ndarray) and not r.flags.writeable else None     # (only read-only arrays are trusted to be unchanged)
            out[k] = buf[0]
        return out

    def _flatten(self, cp: ControlParams):
        gp, mp = cp.geometrical_params, cp.mechanical_params
        cnv = np.asarray(gp.centroid_node_vectors, dtype=float)
        nbd = len(self.bonds)
        bp = mp.bond_params
        # the spring models read a subset of (k_stretch, k_shear, k_rot, reference_vector) (energy.py:30-67): what a model does not
        # read is passed as 0 / a unit dummy vector and comes back with a zero gradient
        zero = np.zeros(nbd)
        out = {
            "centroid_node_vectors": cnv,
            "reference_vector": self._memo("reference_vector", getattr(bp, "reference_vector", None), lambda: np.ascontiguousarray(np.broadcast_to(
                np.asarray(getattr(bp, "reference_vector", np.array([1.0, 0.0])), dtype=float), (nbd, 2)))),
            "k_bond": self._memo("k_bond", (bp.k_stretch, getattr(bp, "k_shear", None), getattr(bp, "k_rot", None)), lambda: np.stack(
                [_bcast(bp.k_stretch, nbd),
                 _bcast(bp.k_shear, nbd) if self.spec.bond_model in (_b.BOND_LINEARIZED, _b.BOND_NONLINEAR) else zero,
                 _bcast(bp.k_rot, nbd) if self.spec.bond_model != _b.BOND_SIMPLE_SPRING else zero], 1)),
        }
        cached = _FLAT_CACHE.get(id(cnv)) if mp.inertia is None else None
        if cached is not None and cached[0]() is cnv and (cached[1] is self.bonds or np.array_equal(cached[1], self.bonds)) and np.array_equal(cached[2], mp.density):
            out["inertia"] = cached[3]        # same design seen through another solver (multi-input problems): reuse
            if self.spec.contact == _b.CONTACT_ANGLE and cached[4] is not None:
                out["void_angle0"] = cached[4]
        elif mp.inertia is None:   # dynamics.py:157-163
            out["inertia"] = compute_inertia(cnv, mp.density)
        else:
            out["inertia"] = np.asarray(mp.inertia, dtype=float).reshape(self.n_blocks, 3)
        def _damping():
            damping = np.zeros((self.n_blocks, 3))
            if self.damped_blocks is not None:   # loading.py:71-106
                damping[self.damped_blocks] = np.broadcast_to(np.asarray(mp.damping, dtype=float), (len(self.damped_blocks), 3))
            return damping
        out["damping"] = self._memo("damping", mp.damping, _damping)
        if self.spec.contact:
            c = mp.contact_params
            if self.spec.contact == _b.CONTACT_DISTANCE:      # energy.py:397-404: absolute node positions enter
                out["block_centroids"] = np.asarray(gp.block_centroids, dtype=float).reshape(self.n_blocks, 2)
                out.pop("void_angle0", None)
            elif "void_angle0" not in out:
                out["void_angle0"] = void_angles0(cnv, self.bonds)
            out["contact"] = np.array([c.min_angle, c.cutoff_angle, c.k_contact], dtype=float)
        if mp.inertia is None and cached is None and not cnv.flags.writeable:
            # geometry arrays that come out of the design cache are read-only and shared: remember what was derived from them
            import weakref
            key = id(cnv)
            if len(_FLAT_CACHE) > 1024:
                _FLAT_CACHE.clear()
            _FLAT_CACHE[key] = (weakref.ref(cnv, lambda _r, k=key: _FLAT_CACHE.pop(k, None)), self.bonds, np.array(mp.density, copy=True),
                                out["inertia"], out.get("void_angle0"))
        fnp = [f.resolve(cp.constraint_params) for f in self.con_terms] + [f.resolve(cp.loading_params) for f in self.load_terms]
        if fnp:
            out["fn_params"] = np.stack(fnp)
        return out

    def _members(self, control_params):
        cps = list(control_params) if isinstance(control_params, (list, tuple)) and not isinstance(control_params, ControlParams) \
            else [control_params]
        if len(cps) == 1 and self.batch > 1:
            cps = cps * self.batch
        if len(cps) != self.batch:
            raise ValueError(f"expected {self.batch} ControlParams, got {len(cps)}")
        return cps

    def estimate_steps_per_interval(self, flat, timepoints):
        """Fixed grid from a Gershgorin-type bound on the largest natural frequency: h * omega_max <= 0.5."""
        inertia, k, cnv = flat["inertia"], flat["k_bond"], flat["centroid_node_vectors"]
        l02 = (flat["reference_vector"] ** 2).sum(1)
        n_b = np.zeros(self.n_blocks)
        kt = np.zeros(self.n_blocks); kr = np.zeros(self.n_blocks)
        r2 = (cnv ** 2).sum(-1).max(1)
        for end in (0, 1):
            blk = self.bonds[:, end] // self.n_npb
            np.add.at(kt, blk, 2 * (k[:, 0] + k[:, 1]))
            np.add.at(kr, blk, 2 * ((k[:, 0] + k[:, 1]) * r2[blk] + k[:, 2] + k[:, 1] * l02 / 4))
            np.add.at(n_b, blk, 1)
        w2 = max((kt / inertia[:, 0]).max(), (kr / inertia[:, 2]).max())
        dt = 0.5 / np.sqrt(w2)
        span = np.diff(np.asarray(timepoints, dtype=float)).max() if len(timepoints) > 1 else 0.0
        return max(1, int(np.ceil(span / dt)))

    def adaptive_grid(self, state0, timepoints, flats, refine=None):
        """(steps per output interval, step boundaries) frozen from a forward-only adaptive solve of the same problem
        (parameters already on the device): the step boundaries the controller accepted for the member that needed
        the most steps, merged with the output times (the controller steps across them and interpolates; a fixed grid
        has to land on them).  Accepted boundaries closer to an output time than 1 % of the neighbouring step are dropped.

        ``refine = k`` (default: the solver's ``grid_refine``, 1) splits every step of the frozen grid into k equal ones.  Why one
        would: the discrete adjoint differentiates the fixed-grid solve exactly, but at loose tolerances (the paper's atol = 1e-4) that
        solve -- on steps the controller sized for ITS error estimate -- is further from the exact gradient than the reference's
        continuous adjoint, which re-integrates the adjoint equations with its own controller (6.9e-3 against 1.7e-3 on the paper
        lattice; with k = 2 the frozen-grid gradient is the closer one, at twice the steps: python -m tests.adjoint_semantics)."""
        ts = np.asarray(timepoints, dtype=float)
        if len(ts) < 2:
            return np.zeros(0, dtype=np.int32), None
        _, st = self.engine.forward_adaptive(state0, ts, self.rtol, self.atol, max_attempts=self.max_attempts)
        self.adaptive_stats = st
        counts = self.engine.adaptive_step_counts()
        acc = self.engine.adaptive_step_times(int(counts.sum(1).argmax()))
        acc = acc[(acc > ts[0]) & (acc < ts[-1])]
        if len(acc):
            step = np.diff(np.concatenate([[ts[0]], acc]))
            near = np.abs(acc[:, None] - ts[None, :]).min(1)
            acc = acc[near > 0.01 * step]
        grid = np.union1d(ts, acc)
        k = self.grid_refine if refine is None else int(refine)
        if k > 1:
            fr = np.arange(k) / k
            grid = np.concatenate([(grid[:-1, None] + np.diff(grid)[:, None] * fr[None, :]).reshape(-1), grid[-1:]])
            grid[::k] = np.union1d(ts, acc)          # the original boundaries exactly (output times must be hit bit for bit)
        spis = np.array([np.count_nonzero((grid >= a) & (grid < b)) for a, b in zip(ts[:-1], ts[1:])], dtype=np.int32)
        return spis, grid

    # -- solve -----------------------------------------------------------------------------------------
    def prepare(self, control_params):
        """Host side of a solve: ``ControlParams`` -> flattened arrays -> device (``dfx_set_params``).  After it the inputs of
        :meth:`solve_resident` are resident in HBM."""
        cps = self._members(control_params)
        flats = [self._flatten(cp) for cp in cps]
        self.engine.set_params(**self._stack(flats))
        self._prepared = (cps, flats)
        return cps, flats

    def __call__(self, state0, timepoints, control_params, keep_trajectory=False, steps_per_interval=None, step_times=None,
                 want_fields=True):
        self.prepare(control_params)
        fields = self.solve_resident(state0, timepoints, keep_trajectory=keep_trajectory, steps_per_interval=steps_per_interval,
                                     step_times=step_times, want_fields=want_fields)
        if fields is None:
            return None
        # one ControlParams (not a list) and batch 1 -> no leading member axis
        return fields[0] if self.batch == 1 and not isinstance(control_params, (list,)) else fields

    def solve_resident(self, state0, timepoints, keep_trajectory=False, steps_per_interval=None, step_times=None, want_fields=True):
        """The solve on the parameters :meth:`prepare` left on the device; returns the fields with their leading member axis (or None)."""
        cps, flats = self._prepared
        self.solve_count += 1
        spi = steps_per_interval if steps_per_interval is not None else self.steps_per_interval
        if np.ndim(timepoints) == 2:
            # one row of output times per member (same number of outputs and of steps per interval: the members advance in the same
            # launches, each on its own time grid) -- forward inputs whose static phases differ in length, static-tuning problem
            if spi is None:
                raise ValueError("per-member timepoints need steps_per_interval (the adaptive controller chooses one grid per call)")
            s0 = None if state0 is None else np.asarray(state0, dtype=float)
            if s0 is not None and s0.ndim == 3:
                s0 = np.broadcast_to(s0, (self.batch,) + s0.shape)
            fields, stats = self.engine.forward(s0, timepoints, spi, keep_trajectory=keep_trajectory, step_times=step_times, want_fields=want_fields)
            self._last = (cps, flats, np.asarray(timepoints, dtype=float))
            self._last_fields = fields
            self.stats = dict(stats, steps_per_interval=spi, step_times=step_times, step_control="fixed")
            return fields
        state0 = np.asarray(state0, dtype=float)
        if state0.ndim == 3:
            state0 = np.broadcast_to(state0, (self.batch,) + state0.shape)
        if spi is None and not keep_trajectory:
            # reference behaviour: adaptive Dormand-Prince controlled by rtol / atol (dynamics.py:166)
            fields, stats = self.engine.forward_adaptive(state0, timepoints, self.rtol, self.atol, max_attempts=self.max_attempts)
            self._last = None
            self.stats = dict(stats, steps_per_interval=None, step_control="adaptive")
            return fields
        control = "fixed"
        if spi is None and self.grid_refine == 1 and self.engine.can_keep_adaptive and os.environ.get("DFX_ADAPTIVE_RECORDS", "1") != "0":
            # the reference's call, differentiable as it stands: the adaptive pass keeps its accepted steps and the reverse sweep is their
            # exact discrete adjoint, output cotangents entering through the dense output -- one forward pass, one reverse sweep, and the
            # gradient belongs to exactly the fields returned (dfx_forward_adaptive_keep; DFX_ADAPTIVE_RECORDS=0: the frozen grid below)
            try:
                fields, stats = self.engine.forward_adaptive(state0, timepoints, self.rtol, self.atol, max_attempts=self.max_attempts,
                                                             keep_trajectory=True, want_fields=want_fields)
            except RuntimeError as e:
                if "forward_adaptive_keep:" not in str(e):
                    raise
                fields = stats = None        # the kept steps do not fit the device (or this physics keeps the two-pass form): frozen grid below
            if stats is not None:
                self._last = (cps, flats, np.asarray(timepoints, dtype=float))
                self._last_fields = fields
                self.stats = dict(stats, steps_per_interval=None, step_times=None, step_control="adaptive-records")
                return fields
        if spi is None:   # the reverse sweep needs a fixed grid: freeze the one the adaptive controller chooses
            spi, step_times = self.adaptive_grid(state0, timepoints, flats)
            control = "adaptive-grid"
        # want_fields=False: the histories stay on the device (objectives evaluated there do not need them on the host)
        fields, stats = self.engine.forward(state0, timepoints, spi, keep_trajectory=keep_trajectory, step_times=step_times,
                                            want_fields=want_fields)
        self._last = (cps, flats, np.asarray(timepoints, dtype=float))
        self._last_fields = fields
        self.stats = dict(stats, steps_per_interval=spi, step_times=step_times, step_control=control)
        return fields

    # -- reverse mode ------------------------------------------------------------------------------------
    def vjp(self, fields_bar):
        """Gradient of sum(fields_bar * fields) w.r.t. every ControlParams leaf and state0.
        Requires the last call to have used ``keep_trajectory=True``.  Returns (ControlParams tree(s), state0_bar)."""
        if self._last is None:
            raise RuntimeError("vjp: call the solver with keep_trajectory=True first")
        fb = np.asarray(fields_bar, dtype=float)
        if fb.ndim == 4:
            fb = fb[None]
        grads, stats = self.engine.adjoint(fb)
        self.adjoint_stats = stats
        trees, s0 = self._unflatten_grads(grads, fb)
        self._last_state0_bar = s0
        # cotangents on the OUTPUTS of prescribed DOFs feed the constraint parameters directly:
        # fields[k, 0, dof] = c_dof(t_k; p), fields[k, 1, dof] = dc_dof/dt(t_k; p)   (dynamics.py:132-134, 169-182)
        if len(self.constrained_pairs) and self.con_terms:
            cps, flats, ts = self._last
            tl = trees if isinstance(trees, list) else [trees]
            n_con = len(self.constrained_pairs)
            dofs = self.constrained_pairs[:, 0] * 3 + self.constrained_pairs[:, 1]
            for m, (cp, tree) in enumerate(zip(cps, tl)):
                fbm = fb[m].reshape(ts.shape[-1], 2, -1)
                for term in self.con_terms:
                    vec = _bcast(term.vector, n_con)
                    wq = fbm[:, 0, dofs] @ vec          # (T,) weights of g(t_k)
                    wv = fbm[:, 1, dofs] @ vec          # (T,) weights of g'(t_k)
                    if not (np.any(wq) or np.any(wv)):
                        continue
                    p = term.resolve(cp.constraint_params)
                    g5 = np.zeros(_b.DFX_FN_PARAMS)
                    for k, t in enumerate(ts[m] if ts.ndim == 2 else ts):
                        if wq[k]:
                            g5 += wq[k] * term.param_partials(float(t), p, "value")
                        if wv[k]:
                            g5 += wv[k] * term.param_partials(float(t), p, "rate")
                    term.scatter_grad(g5, tree.constraint_params, cp.constraint_params)
        return trees, s0

    def timepoints_vjp(self, fields_bar):
        """Cotangent of ``timepoints`` for the cotangent ``fields_bar`` of the last solve -- what ``jax.grad`` through the reference's
        ``solve_dynamics`` returns for its ``timepoints`` argument (dynamics.py:138-148; jax.experimental.ode._odeint_rev, restated in
        oracle/ref_adjoint.py).  Call after ``vjp(fields_bar)`` (its state0 cotangent is used).  With f = (v, a) the right-hand side on the
        free DOFs and (c', c'') the rates of the prescribed ones:

            ts_bar[i] = fields_bar[i] . d fields[i] / d t_i = g_i . f(y_i, t_i)        i >= 1   (moving a measurement time)
            ts_bar[0] = -lambda(t_0+) . f(y_0, t_0)  (+ g_0 . (c', c'') on prescribed DOFs),  lambda(t_0+) = state0_bar - g_0 on the free DOFs

        (the second line is the closed form of _odeint_rev's ``t0_bar``: d/dt (lambda . f) = lambda . df/dt between outputs and lambda jumps by
        g_i at t_i, so  -sum_i g_i . f_i + int lambda . df/dt dt = -lambda(t_0+) . f_0).  These are the derivatives of the CONTINUOUS solution:
        on a fixed grid they differ from the derivative of the discrete map by the integrator's truncation error, like the reference's own.
        One right-hand-side evaluation per output time (the engine's ``dfx_rhs`` hook).  Returns (T,) or (batch, T)."""
        if self._last is None or getattr(self, "_last_fields", None) is None or getattr(self, "_last_state0_bar", None) is None:
            raise RuntimeError("timepoints_vjp: solve with keep_trajectory=True (fields returned), then call vjp(fields_bar) first")
        cps, flats, ts = self._last
        B, nb = self.batch, self.n_blocks
        fb = np.asarray(fields_bar, dtype=float).reshape(B, -1, 2, nb * 3)
        fields = np.asarray(self._last_fields, dtype=float).reshape(B, -1, 2, nb * 3)
        s0b = np.asarray(self._last_state0_bar, dtype=float).reshape(B, 2, nb * 3)
        T = fields.shape[1]
        tsm = np.broadcast_to(ts, (B, T)) if ts.ndim == 1 else ts
        con = self.constrained_pairs[:, 0] * 3 + self.constrained_pairs[:, 1] if len(self.constrained_pairs) else np.zeros(0, dtype=np.int64)
        n_con = len(con)
        out = np.zeros((B, T))
        for i in range(T):
            # members that share this output time are evaluated in one call (one time for all of them unless timepoints has a row per member)
            for t in np.unique(tsm[:, i]):
                rows = np.nonzero(tsm[:, i] == t)[0]
                dy = self.engine.rhs(fields[:, i].reshape(B, 2, nb, 3), float(t)).reshape(B, 2, nb * 3)
                for m in rows:
                    rate = dy[m].copy()                           # (v, a) on the free DOFs, zeros on the prescribed ones
                    if n_con:
                        rate[0, con] = fields[m, i, 1, con]       # c'(t): the prescribed velocity is what the output holds
                        acc = np.zeros(n_con)
                        for term in self.con_terms:
                            p = term.resolve(cps[m].constraint_params)
                            h = 1e-4 * max(float(np.ptp(tsm[m])), 1e-300)      # (central difference of the rate, in the scale of the horizon)
                            acc += _bcast(term.vector, n_con) * (term.rate(float(t) + h, p) - term.rate(float(t) - h, p)) / (2 * h)
                        rate[1, con] = acc                        # c''(t)
                    g = fb[m, i]
                    if i == 0:
                        lam = s0b[m] - g
                        lam[:, con] = 0.0
                        out[m, 0] = -float(np.vdot(lam, rate))
                        if n_con:
                            out[m, 0] += float(np.vdot(g[:, con], rate[:, con]))
                    else:
                        out[m, i] = float(np.vdot(g, rate))
        return out[0] if (B == 1 and ts.ndim == 1) else out

    def vjp_raw(self, fields_bar, which=("centroid_node_vectors", "void_angle0", "inertia")):
        """Reverse sweep for a cotangent of the fields, returning the engine's raw gradient arrays (batch-leading) of the requested
        parameter groups only -- by default what a design reaches.  Asking for every ControlParams leaf (``vjp``) runs the reverse
        stage in its build that also accumulates per-ligament and damping gradients, at up to twice the time per launch."""
        if self._last is None:
            raise RuntimeError("vjp_raw: call the solver with keep_trajectory=True first")
        fb = np.asarray(fields_bar, dtype=float)
        if fb.ndim == 4:
            fb = fb[None]
        which = tuple(w for w in which if not (w == "void_angle0" and self.spec.contact != _b.CONTACT_ANGLE))
        if self.spec.contact == _b.CONTACT_DISTANCE and "block_centroids" not in which:
            which = which + ("block_centroids",)
        grads, stats = self.engine.adjoint(fb, which=which)
        self.adjoint_stats = stats
        return grads

    def kinetic_energy_value_and_vjp(self, target_blocks):
        """objective = sum_t sum_{b in target} m v^2/2 (energy.py:494-499 over problems/quads_focusing.py:461-467),
        evaluated and differentiated on the device (one engine call: the objective rides along with the reverse sweep)."""
        obj, grads, stats = self.engine.kinetic_value_and_grad(target_blocks)
        self.adjoint_stats = stats
        grads = {k: np.array(v) for k, v in grads.items()}      # the gradient tree outlives the engine's result area
        trees, s0 = self._unflatten_grads(grads, None)
        return (obj[0] if self.batch == 1 else obj), trees, s0

    def kinetic_energy_value_and_raw(self, target_blocks, which=("centroid_node_vectors", "void_angle0", "inertia")):
        """Same objective; returns the engine's raw gradient arrays (batch-leading) for the requested parameter groups only:
        read-only views of the engine's pinned result area, valid until the next call on this solver.
        The maps from these to a design (void-angle and inertia chain rules, lattice map) are linear in the cotangent, so a
        caller that sums several solves of ONE design (multi-input problems) applies them once to the sum."""
        which = tuple(w for w in which if not (w == "void_angle0" and self.spec.contact != _b.CONTACT_ANGLE))
        if self.spec.contact == _b.CONTACT_DISTANCE and "block_centroids" not in which:
            which = which + ("block_centroids",)
        obj, grads, stats = self.engine.kinetic_value_and_grad(target_blocks, which=which)
        self.adjoint_stats = stats
        return obj, grads

    @staticmethod
    def _bond_params_bar(bp, kb, refv_bar, like):
        """Gradient tree with the structure of the bond parameters that were passed in (LigamentParams /
        StretchingTorsionalSpringParams / any NamedTuple with a subset of their fields)."""
        cols = {"k_stretch": 0, "k_shear": 1, "k_rot": 2}
        vals = {}
        for name in bp._fields:
            if name in cols:
                vals[name] = like(getattr(bp, name), kb[:, cols[name]])
            elif name == "reference_vector":
                vals[name] = refv_bar
            else:
                vals[name] = None
        return type(bp)(**vals)

    def _unflatten_grads(self, g, fields_bar):
        cps, flats, ts = self._last
        trees = []
        for m, (cp, flat) in enumerate(zip(cps, flats)):
            gp, mp = cp.geometrical_params, cp.mechanical_params
            cnv = flat["centroid_node_vectors"]
            cnv_bar = g["centroid_node_vectors"][m].copy()
            if self.spec.contact == _b.CONTACT_ANGLE:
                cnv_bar += void_angles0_vjp(cnv, self.bonds, g["void_angle0"][m])
            density_bar, inertia_bar = None, None
            if mp.inertia is None:
                vb, density_bar = compute_inertia_vjp(cnv, mp.density, g["inertia"][m])
                cnv_bar += vb
            else:
                inertia_bar = g["inertia"][m].reshape(np.shape(mp.inertia))
            kb = g["k_bond"][m]

            def like(x, full):
                return full.sum() if np.ndim(x) == 0 else full
            damping_bar = 0.0
            if self.damped_blocks is not None:
                rows = g["damping"][m][self.damped_blocks]
                damping_bar = rows.sum() if np.ndim(mp.damping) == 0 else rows.reshape(np.shape(mp.damping)) \
                    if np.shape(mp.damping) == rows.shape else rows.sum(0)
            contact_bar = None
            if self.spec.contact:
                c = g["contact"][m]
                contact_bar = ContactParams(min_angle=c[0], cutoff_angle=c[1], k_contact=c[2])
            con_bar, load_bar = {}, {}
            for f, term in enumerate(self.con_terms):
                term.scatter_grad(g["fn_params"][m][f], con_bar, cp.constraint_params)
            for f, term in enumerate(self.load_terms):
                term.scatter_grad(g["fn_params"][m][len(self.con_terms) + f], load_bar, cp.loading_params)
            trees.append(ControlParams(
                geometrical_params=GeometricalParams(
                    block_centroids=(np.array(g["block_centroids"][m]).reshape(np.shape(gp.block_centroids)) if "block_centroids" in g
                                     else np.zeros_like(np.asarray(gp.block_centroids, dtype=float))),
                    centroid_node_vectors=cnv_bar),
                mechanical_params=MechanicalParams(
                    bond_params=self._bond_params_bar(mp.bond_params, kb, g["reference_vector"][m], like),
                    density=density_bar, inertia=inertia_bar, damping=damping_bar, contact_params=contact_bar),
                loading_params=load_bar, constraint_params=con_bar))
        state0_bar = g["state0"]
        if self.batch == 1:
            return trees[0], state0_bar[0]
        return trees, state0_bar


def setup_dynamic_solver(geometry, energy_fn, loaded_block_DOF_pairs=None, loading_fn=None,
                         constrained_block_DOF_pairs=np.array([]), constrained_DOFs_fn=zero,
                         damped_blocks=None, rtol: float = 1e-8, atol: float = 1e-8, *,
                         integrator: str = "dopri5", steps_per_interval: Optional[int] = None,
                         batch: int = 1, device: int = 0, streams: int = 0, grid_refine: int = 1, _lib=None):
    """Same positional/keyword arguments as ``difflexmm.dynamics.setup_dynamic_solver`` (dynamics.py:60-69).
    Keyword-only extras of this engine: ``steps_per_interval`` (fixed grid), ``batch``, ``device``, ``streams``, and ``grid_refine``:
    with ``keep_trajectory=True`` and no explicit grid the adaptive controller's accepted steps are frozen into the grid the reverse
    sweep differentiates; ``grid_refine = k`` splits each of them into k (accuracy knob of the gradient, see ``adaptive_grid``)."""
    return DynamicSolver(geometry, energy_fn, loaded_block_DOF_pairs, loading_fn, constrained_block_DOF_pairs,
                         constrained_DOFs_fn, damped_blocks, rtol, atol, integrator, steps_per_interval, batch, device, _lib, streams, grid_refine)


def linear_mode_analysis(displacement, geometry, energy_fn, control_params: ControlParams,
                         constrained_block_DOF_pairs=np.array([]), *, device: int = 0, _lib=None, return_stiffness=False):
    """Eigenvalues (squared angular frequencies) and eigenmodes of ``K q = w^2 M q`` around ``displacement`` -- same arguments and
    results as ``difflexmm.dynamics.linear_mode_analysis`` (dynamics.py:189-245): ``(eigenvalues (n_free,), modes (n_free, n_blocks, 3))``,
    every mode scaled to unit Euclidean norm, constrained DOFs zero.

    The reference takes ``jax.hessian`` of the constrained energy; here the stiffness matrix is assembled on the device from the
    engine's Hessian-vector hook (``dfx_rhs_vjp``): with no damping and no loading the acceleration is ``a = -M^-1 grad E``, so the
    cotangent ``e_j`` on acceleration ``j`` returns ``-K[j, :] / m_j`` in the position part -- ``batch`` rows per call.  The dense
    generalised eigenproblem then goes to ``scipy.linalg.eigh`` on the host, as in the reference."""
    import scipy.linalg
    pairs = np.asarray(constrained_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
    n = geometry.n_blocks
    free, _, _ = DOFsInfo(n, pairs)
    nf = len(free)
    B = int(min(64, max(nf, 1)))
    solver = setup_dynamic_solver(geometry, energy_fn, constrained_block_DOF_pairs=pairs, batch=B, device=device, _lib=_lib)
    flat = solver._flatten(control_params)
    solver.engine.set_params(**{k: np.broadcast_to(v, (B,) + np.shape(v)).copy() for k, v in flat.items()})
    inertia = np.asarray(flat["inertia"], dtype=float).reshape(-1)
    u = np.asarray(displacement, dtype=float).reshape(n, 3)
    y = np.zeros((B, 2, n, 3))
    y[:, 0] = u
    y[:, 0].reshape(B, -1)[:, solver.constrained_DOF_ids] = 0.0          # the constrained kinematics hold them at c(0) = 0
    K = np.zeros((nf, nf))
    for j0 in range(0, nf, B):
        rows = free[j0:j0 + B]
        lam = np.zeros((B, 2, n, 3))
        lam[:, 1].reshape(B, -1)[np.arange(len(rows)), rows] = 1.0
        y_bar, _ = solver.engine.rhs_vjp(y, 0.0, lam, which=())
        K[j0:j0 + len(rows)] = -inertia[rows, None] * y_bar[:len(rows), 0].reshape(len(rows), -1)[:, free]
    K = 0.5 * (K + K.T)
    eigenvalues, vectors = scipy.linalg.eigh(K, np.diag(inertia[free]))
    vectors = (vectors / np.linalg.norm(vectors, axis=0)).T                # row-wise, unit norm
    modes = np.zeros((nf, n * 3))
    modes[:, free] = vectors
    out = (eigenvalues, modes.reshape(nf, n, 3))
    return out + (K,) if return_stiffness else out
