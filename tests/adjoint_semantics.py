"""What does jax.grad return in the reference, and how far is it -- and the engine's discrete adjoint -- from the exact gradient?

The reference differentiates ``solve_dynamics`` with the CONTINUOUS adjoint of jax.experimental.ode (oracle/ref_adjoint.py restates
it).  This module feeds that restatement with the right-hand side and its vector-Jacobian products from the engine's C-ABI test hooks
(``dfx_rhs`` / ``dfx_rhs_vjp``: the CPU port here, so the paper-size lattice is affordable; the torch RHS is the independent provider
for small lattices, tests/test_oracle_adjoint.py) and compares, for the target-kinetic-energy objective of
``problems/quads_focusing.py:447-467``:

  (a) the continuous adjoint at the tolerances the reference's problems use (rtol 1e-8, atol 1e-4: quads_focusing notebooks),
  (b) the engine's discrete adjoint on the grid its adaptive controller freezes at the same tolerances,
  (c) both at tight tolerances -- which agree with each other and serve as the exact gradient.

    python -m tests.adjoint_semantics [n1 n2 [n_timepoints]]      (default 24 16 11: the paper lattice, 10 output intervals)

Test infrastructure (imports oracle/); prints the table DESIGN.md section 5 quotes.
"""
import math
import os
import sys
import time

import numpy as np

from difflexmm_amd import problems as P
from difflexmm_amd.geometry import compute_inertia, compute_inertia_vjp, void_angles0_vjp
from difflexmm_amd.utils import ControlParams
from oracle import ref_adjoint as RA
from oracle import ref_ode


class EngineRHS:
    """``func`` / ``vjp`` for ``ref_adjoint.odeint_rev`` from an engine's test hooks.  The augmented state carries one entry per
    leaf of the reference's ``odeint(rhs, state0, timepoints, control_params, inertia)`` arguments: block centroids, node vectors,
    the three stiffnesses, reference vectors, density, damping, contact constants, constraint parameters, reduced inertia."""

    def __init__(self, solver, cp):
        self.s, self.eng = solver, solver.engine
        self.nb, self.free = solver.n_blocks, np.asarray(solver.free_DOF_ids)
        self.n_free = len(self.free)
        cnv = np.asarray(cp.geometrical_params.centroid_node_vectors, dtype=float)
        mp = cp.mechanical_params
        inertia = compute_inertia(cnv, mp.density)
        self.cp = cp._replace(mechanical_params=mp._replace(inertia=inertia))
        flat = solver._flatten(self.cp)
        self.eng.set_params(**{k: v[None] for k, v in flat.items()})
        self.cnv, self.con_names = cnv, sorted(cp.constraint_params)
        nbd = len(solver.bonds)
        self.layout = [("block_centroids", self.nb * 2), ("centroid_node_vectors", cnv.size), ("k", 3), ("reference_vector", nbd * 2),
                       ("density", 1), ("damping", self.nb * 3), ("contact", 3), ("constraint", len(self.con_names)), ("inertia", self.n_free)]
        self.args_size = sum(n for _, n in self.layout)
        self.evals = 0

    def _full(self, y):
        full = np.zeros((2, self.nb * 3))
        full[:, self.free] = np.asarray(y).reshape(2, self.n_free)
        return full.reshape(1, 2, self.nb, 3)

    def func(self, y, t):
        self.evals += 1
        return self.eng.rhs(self._full(y), t)[0].reshape(2, -1)[:, self.free].reshape(-1)

    def vjp(self, y, t, y_bar):
        self.evals += 1
        yb, g = self.eng.rhs_vjp(self._full(y), t, self._full(y_bar))
        vy = yb[0].reshape(2, -1)[:, self.free].reshape(-1)
        # y_bar . df/dt: only the time functions depend on t explicitly -- central difference of the RHS (one entry of the augmented state)
        h = 1e-7 * max(abs(t), 1e-3)
        vt = float(np.dot((self.func(y, t + h) - self.func(y, t - h)) / (2 * h), y_bar))
        cnv_bar = np.array(g["centroid_node_vectors"][0])
        if "void_angle0" in g:
            cnv_bar = cnv_bar + void_angles0_vjp(self.cnv, self.s.bonds, g["void_angle0"][0])
        con = {}
        for f, term in enumerate(self.s.con_terms):
            term.scatter_grad(g["fn_params"][0][f], con, self.cp.constraint_params)
        parts = [np.zeros(self.nb * 2), cnv_bar.reshape(-1), g["k_bond"][0].sum(0), g["reference_vector"][0].reshape(-1), np.zeros(1),
                 g["damping"][0].reshape(-1), g["contact"][0].reshape(-1) if "contact" in g else np.zeros(3),
                 np.array([con.get(n, 0.0) for n in self.con_names]), g["inertia"][0].reshape(-1)[self.free]]
        return vy, vt, np.concatenate(parts)

    def split(self, args_bar):
        out, k = {}, 0
        for name, n in self.layout:
            out[name] = args_bar[k:k + n]
            k += n
        return out


def continuous_adjoint_design_gradient(fw, design, target_blocks, rtol, atol):
    """Objective and design gradient as the reference's jit(value_and_grad(objective)) produces them: forward odeint, reverse
    _odeint_rev, then the chain rules outside the solver (inertia and geometry)."""
    sd = fw.solve_dynamics
    cp = fw.control_params(design)
    prov = EngineRHS(sd, cp)
    nf, free = prov.n_free, list(prov.free)
    st_f, st_r = {}, {}
    ys = ref_ode.odeint(prov.func, np.zeros(2 * nf), fw.timepoints, rtol=rtol, atol=atol, stats=st_f)
    inertia = compute_inertia(prov.cnv, fw.density)
    g = np.zeros_like(ys)
    inertia_bar = np.zeros_like(inertia)
    value = 0.0
    for b in target_blocks:
        for d in range(3):
            col = nf + free.index(int(b) * 3 + d)
            g[:, col] = inertia[b, d] * ys[:, col]
            inertia_bar[b, d] += 0.5 * (ys[:, col] ** 2).sum()
            value += 0.5 * inertia[b, d] * (ys[:, col] ** 2).sum()
    _, _, args_bar = RA.odeint_rev(prov.func, prov.vjp, ys, fw.timepoints, g, prov.args_size, rtol=rtol, atol=atol, stats=st_r)
    bars = prov.split(args_bar)
    flat_in = inertia_bar.reshape(-1)
    flat_in[prov.free] += bars["inertia"]
    cnv_bar = bars["centroid_node_vectors"].reshape(prov.cnv.shape) + compute_inertia_vjp(prov.cnv, fw.density, flat_in.reshape(-1, 3))[0]
    grad = fw.geometry.vjp(design, cnv_bar, bars["block_centroids"].reshape(-1, 2))
    return value, grad, dict(forward=st_f, reverse=st_r, rhs_evals=prov.evals)


def paper_problem(n1, n2, n_timepoints, rtol, atol, lib, grid_refine=1):
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((n1 * n2, 1))
    fw = P.QuadsFocusingForward(n1_blocks=n1, n2_blocks=n2, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5,
                                density=6.18e-9, damping=damping, amplitude=7.5, loading_rate=30.0, input_delay=0.1 / 30, n_excited_blocks=2,
                                loaded_side="left", input_shift=0, simulation_time=2.0 / 30, n_timepoints=n_timepoints, use_contact=True,
                                k_contact=1.5, min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180, rtol=rtol, atol=atol, grid_refine=grid_refine, _lib=lib)
    fw.setup()
    return fw


def relerr(a, b):
    return max(float(np.abs(x - y).max()) for x, y in zip(a, b)) / max(float(np.abs(y).max()) for y in b)


def main(n1=24, n2=16, n_timepoints=11):
    from oracle.cpu import load
    lib = load()
    rng = np.random.default_rng(1000)
    rows = []
    grads = {}
    records = {}
    for label, (rtol, atol) in (("paper", (1e-8, 1e-4)), ("default", (1e-8, 1e-8)), ("tight", (1e-10, 1e-10))):
        fw = paper_problem(n1, n2, n_timepoints, rtol, atol, lib)
        if not grads:
            base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
            design = tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base)
        obj = P.TargetKineticEnergy(fw, (2, 2), (n1 // 6, n2 // 5))
        # round 6: the engine's default -- ONE adaptive pass that keeps its accepted steps, dense-output discrete adjoint
        t0 = time.time()
        v_r, g_r = obj.value_and_grad(design)
        records[label] = (v_r, g_r, int(fw.solve_dynamics.adjoint_stats["steps"]), time.time() - t0)
        os.environ["DFX_ADAPTIVE_RECORDS"] = "0"
        t0 = time.time()
        v_d, g_d = obj.value_and_grad(design)                       # engine: adaptive forward, frozen grid, discrete adjoint
        t_d = time.time() - t0
        os.environ.pop("DFX_ADAPTIVE_RECORDS")
        st = fw.solve_dynamics.stats
        t0 = time.time()
        v_c, g_c, st_c = continuous_adjoint_design_gradient(fw, design, obj.target_blocks, rtol, atol)
        t_c = time.time() - t0
        grads[label] = (v_d, g_d, v_c, g_c)
        rows.append((label, rtol, atol, int(np.sum(st["steps_per_interval"])), st_c["forward"]["accepted"], st_c["reverse"]["accepted"], t_d, t_c))
        fw.solve_dynamics.engine.close()
    # the accuracy knob of the discrete adjoint at the paper's tolerances: every step of the frozen grid split into k (grid_refine)
    refined = []
    for k in (2, 4):
        fw = paper_problem(n1, n2, n_timepoints, 1e-8, 1e-4, lib, grid_refine=k)
        obj = P.TargetKineticEnergy(fw, (2, 2), (n1 // 6, n2 // 5))
        t0 = time.time()
        v_k, g_k = obj.value_and_grad(design)
        refined.append((k, int(np.sum(fw.solve_dynamics.stats["steps_per_interval"])), v_k, g_k, time.time() - t0))
        fw.solve_dynamics.engine.close()
    v_x, g_x = grads["tight"][0], grads["tight"][1]
    print(f"{n1}x{n2} quads, paper constants, {n_timepoints} outputs over 2/f; objective {v_x:.6e}; |grad|_max {max(np.abs(a).max() for a in g_x):.3e}")
    print(f"tight tolerances: continuous vs discrete adjoint {relerr(grads['tight'][3], g_x):.1e} (objective {abs(grads['tight'][2] - v_x) / abs(v_x):.1e})")
    print("tolerances            | discrete adjoint (engine, frozen adaptive grid)      | continuous adjoint (jax semantics, oracle restatement)")
    for label, rtol, atol, n_d, n_f, n_r, t_d, t_c in rows:
        v_d, g_d, v_c, g_c = grads[label]
        print(f"rtol {rtol:.0e} atol {atol:.0e} | steps {n_d:6d}  objective err {abs(v_d - v_x) / abs(v_x):.1e}  gradient err {relerr(g_d, g_x):.1e}  ({t_d:.0f} s) "
              f"| steps {n_f:6d} + {n_r:6d}  objective err {abs(v_c - v_x) / abs(v_x):.1e}  gradient err {relerr(g_c, g_x):.1e}  ({t_c:.0f} s) "
              f"| discrete vs continuous {relerr(g_d, g_c):.1e}")
    for label, (v_r, g_r, n_r, t_r) in records.items():
        print(f"{label:8s} adaptive pass's own records + dense-output adjoint (default since round 6) | steps {n_r:6d}  objective err "
              f"{abs(v_r - v_x) / abs(v_x):.1e}  gradient err {relerr(g_r, g_x):.1e}  ({t_r:.0f} s) | vs continuous adjoint at the same tolerances "
              f"{relerr(g_r, grads[label][3]):.1e} | vs frozen grid {relerr(g_r, grads[label][1]):.1e}")
    for k, n_k, v_k, g_k, t_k in refined:
        print(f"rtol 1e-08 atol 1e-04, grid_refine = {k} | steps {n_k:6d}  objective err {abs(v_k - v_x) / abs(v_x):.1e}  gradient err {relerr(g_k, g_x):.1e}  ({t_k:.0f} s)")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
