"""Generates tests/golden/*.npz from the oracle (oracle/ref_*.py: torch-fp64 + autograd restatement of the reference).

The reference itself cannot be imported in the build container (JAX absent) and ships no golden arrays, so these
fixtures are outputs of the oracle restatement on seeded inputs; they freeze it (any later edit of the oracle or of
the engine shows up as a diff) and travel to the GPU box.  Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_dynamics as OD  # noqa: E402
from oracle.cpu import load  # noqa: E402
from tests.common import Case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T64 = lambda x, g=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=g)  # noqa: E731


def rhs_cases():
    out = {}
    for lattice, n, cut in (("quads", 4, 42.0), ("kagome", 3, 125.0)):
        for nonlinear in (True, False):
            c = Case(lattice, n, nonlinear, True, seed=7, lib=load(), cutoff_deg=cut)
            y = c.random_state()
            osol = c.oracle_solver()
            free = osol.free_DOF_ids
            r = osol.rhs(T64(y.reshape(2, -1)[:, free]), 0.012, c.oracle_cp(), osol.reduced_inertia(c.oracle_cp()))
            key = f"{lattice}{n}_{'nl' if nonlinear else 'lin'}"
            out[key + "_y"] = y
            out[key + "_dy_free"] = r.detach().numpy()
            out[key + "_free"] = free
    np.savez_compressed(os.path.join(OUT, "rhs.npz"), **out)


def adaptive_trajectory():
    """C1-like: 8x8 quads, linearised ligaments, no contact, no damping; adaptive Dopri5 (reference odeint semantics)."""
    c = Case("quads", 8, False, False, damping=False, seed=1, lib=load())
    fast = dict(loading_rate=T64(600.0), input_delay=T64(1e-4))
    ts = np.linspace(0.0, 4e-3, 11)
    osol = c.oracle_solver(rtol=1e-8, atol=1e-8)
    f = osol(np.zeros((2, 64, 3)), ts, c.oracle_cp(fast)).numpy()
    np.savez_compressed(os.path.join(OUT, "adaptive_8x8.npz"), timepoints=ts, fields=f, attempted=osol.stats["attempted"],
                        accepted=osol.stats["accepted"], loading_rate=600.0, input_delay=1e-4)


def focusing_gradient():
    """6x6 quads focusing problem: objective = target kinetic energy, gradient w.r.t. the design by autograd through the
    unrolled fixed-grid oracle (the oracle's stand-in for jax.grad through solve_dynamics)."""
    from oracle import ref_energy as OE, ref_geometry as OG
    c = Case("quads", 6, True, True, seed=9, lib=load(), cutoff_deg=42.0)
    lv = dict(loading_rate=T64(3000.0), input_delay=T64(1e-5))
    ts = np.linspace(0.0, 4e-4, 5)
    spi = 8
    design = [T64(d, True) for d in c.design]
    cnv = c.ogeo.centroid_node_vectors(*design)
    cen = c.ogeo.block_centroids(*design)
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=spi)
    hist, _ = OD.solve_fixed_differentiable(osol, c.ogeo, T64(np.zeros((2, 36, 3))), ts, c.oracle_cp(dict(cnv=cnv, cen=cen, **lv)), spi)
    target = np.array([14, 15, 20, 21])
    free = list(osol.free_DOF_ids)
    inertia = OG.compute_inertia(cnv, 6.18e-9)
    obj = 0.0
    for b in target:
        for d in range(3):
            col = free.index(b * 3 + d)
            obj = obj + (inertia[b, d] * hist[:, 1, col] ** 2 / 2).sum()
    g = torch.autograd.grad(obj, design)
    np.savez_compressed(os.path.join(OUT, "focusing_6x6.npz"), timepoints=ts, spi=spi, target=target, objective=obj.item(),
                        grad_h=g[0].numpy(), grad_v=g[1].numpy(), design_h=c.design[0], design_v=c.design[1])


PAPER = dict(spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9, amplitude=7.5,
             n_excited_blocks=2, use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180)


def paper_damping(n_blocks, spacing=15.0):
    return 0.0186 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * spacing ** 2 * 1.19)] * 2 +
                             [2 * math.sqrt(0.02175026 * 6.18e-9 * spacing ** 4 * 1.5)]) * np.ones((n_blocks, 1))


def problem_layer():
    """Row a18: the callers.  Index lists of the boundary conditions / targets, pulse values, and objective + design gradient of the
    focusing problems on the PAPER lattices (24x16 quads loaded from each of the four sides, 20x12 kagome), all from
    oracle/ref_problems.py (restatement of problems/quads_focusing.py, kagome_focusing.py, quads_focusing_multi_input.py)."""
    from oracle import ref_problems as RP
    out = {}
    for side, shift in (("left", 0), ("right", -2), ("bottom", -4), ("top", 3)):
        bc = RP.quads_constraints(24, 16, 2, side, shift, 2)
        for k, v in bc.items():
            out[f"quads_{side}_{k}"] = v
    out["quads_target_2x2_4_3"] = RP.quads_target_blocks(24, 16, (2, 2), (4, 3))
    out["quads_target_3x2_m5_2"] = RP.quads_target_blocks(24, 16, (3, 2), (-5, 2))
    bc = RP.kagome_constraints(20, 12, 2, 2)
    for k, v in bc.items():
        out[f"kagome_{k}"] = v
    out["kagome_target_2x2_3_3"] = RP.kagome_target_blocks(20, 12, (2, 2), (3, 3))
    tt = np.linspace(-1e-3, 0.05, 41)
    out["pulse_t"] = tt
    out["pulse_values"] = np.array([float(RP.pulse(t, 7.5, 30.0)) for t in tt])
    fn = RP.make_constrained_DOFs_fn(out["quads_left_constrained_DOFs_loading_vector"])
    out["constrained_fn_t"] = np.array([0.004, 0.02])
    out["constrained_fn_values"] = np.stack([fn(t, 7.5, 30.0, 0.1 / 30.0).numpy() for t in (0.004, 0.02)])
    np.savez_compressed(os.path.join(OUT, "problems_bc.npz"), **out)

    # objective + gradient, short window with a fast pulse (the wave reaches the target inside it), fixed grid
    spi = 6
    rng = np.random.default_rng(1000)
    res = {}
    fast = dict(loading_rate=300.0, input_delay=1e-4, simulation_time=6e-4, n_timepoints=4)
    probs = [RP.ForwardProblem("quads", 24, 16, damping=paper_damping(384), loaded_side=s, input_shift=sh, **PAPER, **fast)
             for s, sh in (("left", 0), ("right", -2), ("bottom", -4))]
    base = probs[0].geometry.get_design_from_rotated_square(25 * math.pi / 180)
    design_np = tuple(np.asarray(b) + rng.uniform(-0.3, 0.3, np.shape(b)) for b in base)
    design = [T64(d, True) for d in design_np]
    target = RP.quads_target_blocks(24, 16, (2, 2), (-10, 0))      # next to the left drive: reached within the window
    total, vals = RP.multi_input_objective(probs, design, target, [1.0, 0.5, 0.25], spi)
    g = torch.autograd.grad(total, design)
    res.update(quads_design_h=design_np[0], quads_design_v=design_np[1], quads_target=target, quads_weights=np.array([1.0, 0.5, 0.25]),
               quads_individual=vals.detach().numpy(), quads_objective=total.item(), quads_grad_h=g[0].numpy(), quads_grad_v=g[1].numpy(),
               spi=spi, **{k: np.float64(v) for k, v in fast.items()})
    kp = RP.ForwardProblem("kagome", 20, 12, damping=paper_damping(480, 20.0), **dict(PAPER, spacing=20.0), **fast)
    from difflexmm_amd.geometry import KagomeGeometry
    shapes = KagomeGeometry(20, 12, 20.0 * np.array([[1.0, 0.0], [0.5, math.sqrt(3) / 2]]), 2.25).design_shapes()
    kdesign_np = tuple(rng.uniform(-0.3, 0.3, sh) for sh in shapes)
    kdesign = [T64(d, True) for d in kdesign_np]
    ktarget = RP.kagome_target_blocks(20, 12, (2, 2), (-8, 0))
    kval = RP.target_kinetic_energy(kp, kdesign, ktarget, spi)
    kg = torch.autograd.grad(kval, kdesign)
    res.update(kagome_target=ktarget, kagome_objective=kval.item(), **{f"kagome_design_{i}": d for i, d in enumerate(kdesign_np)},
               **{f"kagome_grad_{i}": a.numpy() for i, a in enumerate(kg)})
    np.savez_compressed(os.path.join(OUT, "problems_objective.npz"), **res)


def long_horizon():
    """About a thousand steps on the autograd tape (the horizon is bounded by the dynamics, not by the tape: with the contact engaged at
    these amplitudes a rounding difference grows ~100x per 400 steps -- oracle against C++ port 5e-16 / 9e-14 / 2e-12 / 3e-9 after
    400 / 800 / 1200 / 2000 steps on the 8x8 quads, 5e-14 / 3e-12 / 2e-10 after 400 / 800 / 1200 on the 4x4 kagome): trajectory, objective and design gradient of small lattices over a horizon in which the
    pulse crosses the lattice and rings down (nonlinear ligaments + damping + angle contact ENGAGED: the cutoff sits above the undeformed
    void angles, `contact_energy` at the output times is in the file).  Pins the long-horizon behaviour of the engine and of the C++ port
    (the second checker) to the torch oracle; the other oracle trajectories are a few tens of steps."""
    from oracle import ref_energy as OE, ref_geometry as OG
    for lattice, n, cut, steps, seed in (("quads", 8, 42.0, 1200, 21), ("kagome", 4, 125.0, 800, 22)):
        c = Case(lattice, n, True, True, seed=seed, lib=load(), cutoff_deg=cut)
        nb = c.geo.n_blocks
        lv = dict(loading_rate=T64(3000.0), input_delay=T64(1e-5))
        ts = np.linspace(0.0, 1.5e-6 * steps, 9)
        spi = steps // 8
        design = [T64(d, True) for d in c.design]
        cnv = c.ogeo.centroid_node_vectors(*design)
        cen = c.ogeo.block_centroids(*design)
        osol = c.oracle_solver(integrator="fixed", steps_per_interval=spi)
        hist, _ = OD.solve_fixed_differentiable(osol, c.ogeo, T64(np.zeros((2, nb, 3))), ts, c.oracle_cp(dict(cnv=cnv, cen=cen, **lv)), spi)
        free = list(osol.free_DOF_ids)
        target = np.array([nb // 2 + 1, nb // 2 + 2])
        inertia = OG.compute_inertia(cnv, 6.18e-9)
        obj = 0.0
        for b in target:
            for d in range(3):
                col = free.index(b * 3 + d)
                obj = obj + (inertia[b, d] * hist[:, 1, col] ** 2 / 2).sum()
        g = torch.autograd.grad(obj, design)
        cp = c.oracle_cp(lv)
        fields = osol(np.zeros((2, nb, 3)), ts, cp).detach()    # all DOFs, prescribed ones with their rate (not under no_grad: the rate is autograd's)
        with torch.no_grad():
            ce = OE.build_contact_energy(c.bonds)
            contact = np.array([float(ce(fields[k, 0], cp)) for k in range(len(ts))])
        for part in (0, 1):       # the tape-free fixed-grid solve of the oracle and the taped one: the same arithmetic
            a, b = fields.reshape(len(ts), 2, -1)[:, part][:, free], hist.detach()[:, part]
            assert float((a - b).abs().max() / b.abs().max()) < 1e-10
        np.savez_compressed(os.path.join(OUT, f"long_horizon_{lattice}.npz"), timepoints=ts, spi=spi, target=target, seed=seed, n=n,
                            cutoff_deg=cut, fields=fields.numpy(), objective=obj.item(), contact_energy=contact,
                            **{f"design_{i}": d for i, d in enumerate(c.design)}, **{f"grad_{i}": a.numpy() for i, a in enumerate(g)})
        print(lattice, "objective", obj.item(), "contact energy", contact[0], "->", contact[-1], flush=True)


def long_horizon_32(steps=2000, n=32, seed=23, cut=42.0, intervals=8, name="long_horizon_quads32", keep=None, dt=1.5e-6):
    """The same at a size where the engine's size-dependent code runs (round-4 verdict #5): 32 x 32 quads (1 024 blocks: 64 waves, several
    workgroups, the XCD-banded order, more than one segment), contact engaged everywhere, 2 000 fixed Dopri5 steps.  Two thousand steps
    of 1 024 blocks do not fit on one autograd tape (33 MB per step), so the gradient is assembled interval by interval: a first pass
    keeps the state at the start of every interval; the reverse pass re-runs ONE interval on the tape from its stored state (a leaf)
    and differentiates  L_k = (objective terms of the interval's output) + <state cotangent of the next interval, final state>  with
    respect to that leaf and the design -- the chain rule written out, still autograd through the unrolled oracle inside an interval."""
    from oracle import ref_energy as OE, ref_geometry as OG
    import time
    c = Case("quads", n, True, True, seed=seed, lib=load(), cutoff_deg=cut)
    nb = c.geo.n_blocks
    lv = dict(loading_rate=T64(3000.0), input_delay=T64(1e-5))
    ts = np.linspace(0.0, dt * steps, intervals + 1)
    spi = steps // intervals
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=spi)
    free = torch.as_tensor(osol.free_DOF_ids, dtype=torch.long)
    free_l = list(osol.free_DOF_ids)
    target = np.array([nb // 2 + n // 2 + 1, nb // 2 + n // 2 + 2])
    cols = [[free_l.index(b * 3 + d) for d in range(3)] for b in target]

    def interval(y_free, k, design):
        cnv = c.ogeo.centroid_node_vectors(*design)
        cen = c.ogeo.block_centroids(*design)
        full = torch.zeros(2, nb * 3, dtype=torch.float64)
        full = full.index_copy(1, free, y_free)
        hist, _ = OD.solve_fixed_differentiable(osol, c.ogeo, full.reshape(2, nb, 3), ts[k:k + 2], c.oracle_cp(dict(cnv=cnv, cen=cen, **lv)), spi)
        inertia = OG.compute_inertia(cnv, 6.18e-9)
        return hist[1], inertia

    def kinetic(y_free, inertia):
        e = 0.0
        for b, cs in zip(target, cols):
            for d in range(3):
                e = e + inertia[b, d] * y_free[1, cs[d]] ** 2 / 2
        return e

    t0 = time.time()
    design0 = [T64(d) for d in c.design]
    states = [torch.zeros(2, len(free_l), dtype=torch.float64)]
    for k in range(intervals):
        y1, _ = interval(states[-1], k, design0)
        states.append(y1.detach())
        print("forward interval", k, time.time() - t0, "s", flush=True)
    grads = [torch.zeros_like(d) for d in design0]
    ybar = torch.zeros_like(states[-1])
    obj = 0.0
    for k in range(intervals - 1, -1, -1):
        design = [T64(d, True) for d in c.design]
        y0 = states[k].clone().requires_grad_(True)
        y1, inertia = interval(y0, k, design)
        e = kinetic(y1, inertia)
        L = e + (ybar * y1).sum()
        g = torch.autograd.grad(L, [y0] + design)
        ybar = g[0]
        for a, b in zip(grads, g[1:]):
            a += b
        obj += e.item()
        print("reverse interval", k, time.time() - t0, "s", flush=True)
    # (output 0 is the state at rest: its kinetic energy and gradient are zero)
    np.savez_compressed(f"/tmp/{name}_partial.npz", states=torch.stack(states).numpy(), objective=obj, **{f"grad_{i}": a.numpy() for i, a in enumerate(grads)})
    cp = c.oracle_cp(lv)
    fields = osol(np.zeros((2, nb, 3)), ts, cp).detach()     # the oracle's ordinary (tape-free) fixed-grid solve: all DOFs, prescribed ones with their rate
    # the two oracle paths restart their first stage differently at an interval boundary (f(y, t_k) against the carried k_7): the same
    # arithmetic up to the last bit of t, and this trajectory amplifies a last-bit difference -- the measured agreement is recorded
    drift = []
    for k in range(intervals + 1):
        d = 0.0
        for part in (0, 1):
            a, b = fields.reshape(len(ts), 2, -1)[k, part][free], states[k][part]
            d = max(d, float((a - b).abs().max() / max(float(b.abs().max()), 1e-300)))
        drift.append(d)
    print("taped (interval by interval) against tape-free oracle solve, per output:", drift, flush=True)
    with torch.no_grad():
        ce = OE.build_contact_energy(c.bonds)
        contact = np.array([float(ce(fields[k, 0], cp)) for k in range(len(ts))])
    keep = np.arange(0, intervals + 1, 2) if keep is None else np.asarray(keep)          # every other output row travels (file size)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), timepoints=ts, spi=spi, target=target, seed=seed, n=n, cutoff_deg=cut,
                        rows=keep, fields=fields.numpy()[keep], objective=obj, contact_energy=contact, oracle_paths_agree=np.array(drift),
                        **{f"design_{i}": d for i, d in enumerate(c.design)}, **{f"grad_{i}": a.numpy() for i, a in enumerate(grads)})
    print(name, "objective", obj, "contact energy", contact[0], "->", contact[-1], "total", time.time() - t0, "s", flush=True)


def pulse_rs_script(n1_cells=20, n2_cells=10, name="pulse_rs"):
    """scripts/pulse_RS.py as written, on the oracle: 20 x 10 cells of rotated squares (40 x 20 blocks) at 0.35 rad, nonlinear
    ligaments (1, 0.33, 0.0075), no constraints, no damping, the sech^2 tanh force pulse on the second column of blocks, the reference's
    default adaptive odeint (rtol = atol = 1e-8) over t = 0 .. n1_blocks; inertia from the geometry.  Kept: every 11th of the script's
    100 output times (the controller's steps do not depend on the output times)."""
    import time
    from oracle import ref_energy as OE, ref_geometry as OG
    g = OG.RotatedSquareGeometry(n1_cells, n2_cells, 1.0, 0.1)
    ang = 0.35
    cnv, cen = g.centroid_node_vectors(ang), g.block_centroids(ang)
    inertia = OG.compute_inertia(cnv, 1.0)
    energy = OE.build_strain_energy(g.bond_connectivity(), OE.ligament_energy)
    loaded = np.array([[g.n1_blocks * i + 1, 0] for i in range(g.n2_blocks)])
    amp, sharp = 0.3, 4.0

    def loading(state, t):
        tt = torch.as_tensor(t, dtype=torch.float64)
        return 2 * amp / sharp ** 2 * torch.cosh(tt / sharp - 3) ** (-2) * torch.tanh(3 - tt / sharp)

    solver = OD.setup_dynamic_solver(g, energy, loaded_block_DOF_pairs=loaded, loading_fn=loading, constrained_block_DOF_pairs=np.zeros((0, 2), dtype=np.int64))
    cp = OE.ControlParams(OE.GeometricalParams(T64(cen), T64(cnv)),
                          OE.MechanicalParams(OE.LigamentParams(T64(1.0), T64(0.33), T64(0.0075), T64(g.reference_bond_vectors())), T64(1.0), inertia, None, None))
    ts = np.linspace(0.0, float(g.n1_blocks), 100)[::11]
    t0 = time.time()
    f = solver(np.zeros((2, g.n_blocks, 3)), ts, cp).numpy()
    print(name, "oracle adaptive solve", time.time() - t0, "s", solver.stats, "max |q|", np.abs(f[:, 0]).max(), flush=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), timepoints=ts, fields=f, attempted=solver.stats["attempted"], accepted=solver.stats["accepted"],
                        n1_cells=n1_cells, n2_cells=n2_cells, inertia=np.asarray(inertia))


def long_horizon_128():
    """BASELINE config 3's lattice at full size (round-5 verdict item 5): 128 x 128 quads (16 384 blocks: 1 024 waves, every XCD band, four
    workgroups per compute unit), contact engaged everywhere, 500 fixed Dopri5 steps in 25 intervals of 20 (a tape of 20 steps of this
    lattice is ~10 GB) -- so that full-size long-horizon parity no longer leans on the C++ port, which shares the engine's physics headers.
    Two output rows travel (the middle one and the last: 786 KB each)."""
    long_horizon_32(steps=500, n=128, seed=29, cut=42.0, intervals=25, name="long_horizon_quads128", keep=[0, 13, 25])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "long_horizon_128":
        long_horizon_128()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "problems":
        problem_layer()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pulse_rs":
        pulse_rs_script()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        long_horizon()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "long32":
        long_horizon_32()
        sys.exit(0)
    rhs_cases()
    adaptive_trajectory()
    focusing_gradient()
    problem_layer()
    long_horizon()
    long_horizon_32()
    pulse_rs_script()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
