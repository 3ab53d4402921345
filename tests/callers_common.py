"""Shared bodies of the parity tests for the remaining callers of the hot path (row a18): the energy-splitting objective
(problems/quads_energy_splitting.py), the restricted design space and the boundary-angle constraint
(problems/quads_focusing_restricted_space.py, quads_focusing.py:473-532), the rotated-squares reference design
(problems/reference_design.py).  Engine under test (CPU port / HIP) against autograd through the unrolled oracle."""
import math

import numpy as np
import torch

from difflexmm_amd import problems as P
from oracle import ref_problems as RP

N1, N2, SPI, NT, TSIM = 7, 6, 8, 4, 6e-4
KW = dict(spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9, amplitude=5.0, loading_rate=2500.0,
          input_delay=2e-5, n_excited_blocks=2, simulation_time=TSIM, n_timepoints=NT, use_contact=True, k_contact=1.5,
          min_angle=5 * math.pi / 180, cutoff_angle=45 * math.pi / 180)


def damping(n=N1 * N2):
    return 0.05 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((n, 1))


def quads_forward(lib):
    fw = P.QuadsFocusingForward(n1_blocks=N1, n2_blocks=N2, damping=damping(), loaded_side="left", input_shift=0, steps_per_interval=SPI,
                                _lib=lib, **KW)
    fw.setup()
    ofw = RP.ForwardProblem("quads", N1, N2, KW["spacing"], KW["bond_length"], KW["k_stretch"], KW["k_shear"], KW["k_rot"], KW["density"],
                            damping(), KW["amplitude"], KW["loading_rate"], KW["input_delay"], 2, TSIM, NT, "left", 0, use_contact=True,
                            k_contact=1.5, min_angle=KW["min_angle"], cutoff_angle=KW["cutoff_angle"])
    rng = np.random.default_rng(19)
    base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
    x = tuple(b + rng.uniform(-0.2, 0.2, b.shape) for b in base)
    return fw, ofw, x


def check_energy_splitting(lib, tol=1e-9):
    """quads_energy_splitting.py:56-88: one input, three targets (two of them overlapping), weights of both signs."""
    fw, ofw, x = quads_forward(lib)
    sizes, shifts, weights = ((2, 2), (1, 2), (2, 1)), ((1, 1), (-1, 0), (1, 0)), (1.0, -0.5, 0.25)
    obj = P.SplitTargetKineticEnergy(fw, sizes, shifts, weights)
    v, g = obj.value_and_grad(x)
    tbs = [RP.quads_target_blocks(N1, N2, s, sh) for s, sh in zip(sizes, shifts)]
    for a, b in zip(tbs, obj.target_blocks_list):
        assert np.array_equal(a, b)
    xt = [torch.tensor(a, requires_grad=True) for a in x]
    vals = RP.split_kinetic_energies(ofw, xt, tbs, SPI)
    ov = (torch.tensor(weights, dtype=torch.float64) * vals).sum()
    og = torch.autograd.grad(ov, xt)
    assert vals.detach().numpy().min() > 0
    assert np.abs(obj.last_individual - vals.detach().numpy()).max() < tol * vals.detach().numpy().max()
    assert abs(v - ov.item()) < tol * abs(ov.item())
    for a, b in zip(g, og):
        assert np.abs(a - b.numpy()).max() < tol * np.abs(b.numpy()).max()
    assert abs(obj.value(x) - v) < 1e-12 * abs(v) and np.allclose(obj.individual(x), obj.last_individual, rtol=1e-12)
    # the loop (quads_energy_splitting.py:121-266) keeps the per-target values of every evaluation; dict round trip
    opt = P.OptimizationProblem(obj, name="quads_energy_splitting")
    opt.run_optimization_nlopt(x, 2, lower_bound=-4.5, upper_bound=4.5, min_void_angle=0.0, min_block_angle=0.0, min_edge_length=1.0, verbose=False)
    assert len(opt.objective_values_individual) == len(opt.objective_values) == 2 and opt.objective_values_individual[0].shape == (3,)
    assert abs(float(np.dot(weights, opt.objective_values_individual[0])) - opt.objective_values[0]) < 1e-12 * abs(opt.objective_values[0])
    d = P.OptimizationProblem.from_dict(opt.to_dict(), _lib=lib)
    assert isinstance(d.objective, P.SplitTargetKineticEnergy) and d.objective.target_shifts == obj.target_shifts
    assert len(d.objective_values_individual) == 2


def check_constraints_against_oracle():
    """Angle constraints with and without the boundary rows, edge-length constraints and their sparse Jacobians against the oracle twin
    (values 1e-13; Jacobians against autograd 1e-11).  Host logic only: no engine involved."""
    from difflexmm_amd.geometry import QuadGeometry
    from oracle import ref_geometry as OG
    g, og = QuadGeometry(N1, N2, 15.0, 2.25), OG.QuadGeometry(N1, N2, 15.0, 2.25)
    rng = np.random.default_rng(3)
    base = g.get_design_from_rotated_square(25 * math.pi / 180)
    x = tuple(b + rng.uniform(-0.4, 0.4, b.shape) for b in base)
    assert np.array_equal(P.quads_boundary_nodes(g), RP.quads_boundary_nodes(N1, N2)) and len(P.quads_boundary_nodes(g)) == 2 * (N1 + N2)
    xt = [torch.tensor(a, requires_grad=True) for a in x]
    flat = torch.cat([a.reshape(-1) for a in xt])
    for boundary in (False, True):
        got = P.angle_constraints(g, x, 0.1, 0.2, boundary)
        ref = RP.angle_constraints(og, xt, 0.1, 0.2, boundary)
        assert got.shape == (4 * len(g.bond_connectivity()) + (2 * (N1 + N2) if boundary else 0),)
        assert np.abs(got - ref.detach().numpy()).max() < 1e-13

        def f(v):
            h = v[:xt[0].numel()].reshape(xt[0].shape)
            w = v[xt[0].numel():].reshape(xt[1].shape)
            return RP.angle_constraints(og, (h, w), 0.1, 0.2, boundary)
        J = torch.autograd.functional.jacobian(f, flat.detach()).numpy()
        assert np.abs(P.angle_constraints_jac(g, x, boundary).toarray() - J).max() < 1e-11 * np.abs(J).max()
    got = P.edge_length_constraints(g, x, 0.5)
    assert np.abs(got - RP.edge_length_constraints(og, xt, 0.5).detach().numpy()).max() < 1e-13


def check_restricted_design_space(lib, tol=1e-9):
    """quads_focusing_restricted_space.py: masks and maps equal to the oracle twin; objective and gradient w.r.t. the REDUCED shifts
    against autograd through the unrolled oracle; two evaluations of the loop leave what lies outside the patch untouched."""
    fw, ofw, x = quads_forward(lib)
    target_size, target_shift, patch = (2, 2), (1, 0), 3
    obj = P.TargetKineticEnergy(fw, target_size, target_shift)
    opt = P.OptimizationProblem(obj, initial_guess_all=x, design_patch_size=patch)
    hm, vm = RP.restricted_space_masks(N1, N2, (x[0].shape, x[1].shape), target_shift, patch)
    assert np.array_equal(hm, opt.space.horizontal_shifts_mask) and np.array_equal(vm, opt.space.vertical_shifts_mask)
    assert 0 < hm.sum() < hm.size and 0 < vm.sum() < vm.size
    red = opt.all_to_reduced_shifts(x)
    rng = np.random.default_rng(5)
    red = tuple(r + rng.uniform(-0.1, 0.1, r.shape) for r in red)
    full = opt.reduced_to_all_shifts(red)
    for a, b, m in zip(full, x, (hm, vm)):
        assert np.array_equal(a[~m], b[~m]) and not np.array_equal(a[m], b[m])
    # gradient w.r.t. the reduced shifts = the masked entries of the full gradient: against autograd through the oracle's own maps
    v, gfull = obj.value_and_grad(full)
    rt = [torch.tensor(r, requires_grad=True) for r in red]
    tb = RP.quads_target_blocks(N1, N2, target_size, target_shift)
    ov = RP.target_kinetic_energy(ofw, RP.reduced_to_all_shifts(rt, x, (hm, vm)), tb, SPI)
    og = torch.autograd.grad(ov, rt)
    assert abs(v - ov.item()) < tol * abs(ov.item())
    gred = opt.all_to_reduced_shifts(gfull)
    for a, b in zip(gred, og):
        assert np.abs(a - b.numpy()).max() < tol * np.abs(b.numpy()).max()
    # the loop on the reduced vector (MMA standing in for NLopt): histories hold reduced shifts, the forward sees full designs
    best = opt.run_optimization_nlopt(opt.all_to_reduced_shifts(x), 2, lower_bound=-4.5, upper_bound=4.5, min_void_angle=0.0,
                                      min_block_angle=0.0, min_edge_length=1.0, boundary_angle_constraint=False, verbose=False)
    assert len(opt.objective_values) == 2 and all(a.shape == b.shape for a, b in zip(opt.design_values[-1], red))
    assert abs(opt.objective_values[0] - obj.value(x)) < 1e-10 * abs(opt.objective_values[0])
    sol = opt.compute_best_forward()
    assert np.isfinite(sol.fields).all() and len(best) == 2
    d = P.OptimizationProblem.from_dict(opt.to_dict(), _lib=lib)
    assert d.space is not None and np.array_equal(d.space.columns, opt.space.columns) and d.design_patch_size == patch


def check_reference_design(lib, tol=1e-10):
    """problems/reference_design.py: rotated squares at an initial angle with the focusing problems' boundary conditions -- boundary
    lists and a short fixed-grid trajectory against the oracle twin; the recorded-signal variant of setup() reproduces the table."""
    n1, n2, angle = 8, 6, 25 * math.pi / 180
    kw = dict(KW, n_excited_blocks=2)
    fw = P.RotatedSquaresForward(n1_blocks=n1, n2_blocks=n2, damping=damping(n1 * n2), loaded_side="right", input_shift=-1,
                                 steps_per_interval=SPI, initial_angle=angle, _lib=lib, **kw)
    fw.setup()
    ofw = RP.ForwardProblem("rotated_squares", n1, n2, kw["spacing"], kw["bond_length"], kw["k_stretch"], kw["k_shear"], kw["k_rot"],
                            kw["density"], damping(n1 * n2), kw["amplitude"], kw["loading_rate"], kw["input_delay"], 2, TSIM, NT, "right", -1,
                            use_contact=True, k_contact=1.5, min_angle=kw["min_angle"], cutoff_angle=kw["cutoff_angle"])
    assert np.array_equal(fw.constrained_block_DOF_pairs, ofw.constrained_block_DOF_pairs)
    assert np.array_equal(fw.driven_blocks_ids, ofw.driven_blocks_ids) and np.array_equal(fw.clamped_blocks_ids, ofw.clamped_blocks_ids)
    assert fw.signed_amplitude == -kw["amplitude"] and fw.name == "rotated_squares"
    sol = fw.solve()
    hist, osolver = ofw.velocity_history((torch.tensor(angle, dtype=torch.float64),), SPI)
    free = osolver.free_DOF_ids
    got = sol.fields.reshape(NT, 2, -1)[:, :, free]
    for part in (0, 1):
        ref = hist[:, part].detach().numpy()
        assert np.abs(ref).max() > 0 and np.abs(got[:, part] - ref).max() < tol * np.abs(ref).max()
    assert np.allclose(sol.centroid_node_vectors, ofw.geometry.centroid_node_vectors(torch.tensor(angle, dtype=torch.float64)).numpy(), rtol=1e-14, atol=1e-14)
    out = fw.compute_response_data()
    assert out["kinetic_energy"].shape == (NT, n1 * n2) and out["strain_energy_stretch"].shape == (NT, len(fw.bond_connectivity))
    # recorded input: the driven DOF follows the table (end values held), the pulse parameters are not used
    from difflexmm_amd import loading as L
    tt = np.linspace(0.0, TSIM / 2, 9)
    fw2 = P.RotatedSquaresForward(n1_blocks=n1, n2_blocks=n2, damping=damping(n1 * n2), loaded_side="left", input_shift=0,
                                  steps_per_interval=SPI, initial_angle=angle, _lib=lib, **kw)
    fw2.setup(excited_blocks_fn=L.Table(tt, 0.3 * np.sin(np.pi * tt / tt[-1]) ** 2 + 0.2 * tt / tt[-1]))
    s2 = fw2.solve()
    expect = np.interp(fw2.timepoints, tt, 0.3 * np.sin(np.pi * tt / tt[-1]) ** 2 + 0.2 * tt / tt[-1])
    assert np.allclose(s2.fields[:, 0, fw2.driven_blocks_ids[0], 0], expect, atol=1e-14)


def check_recorded_input_signal(lib):
    """``setup(excited_blocks_fn=...)`` of the forward problems (problems/quads_focusing.py:213-222, kagome_focusing.py, quads_spin.py):
    the driven DOFs follow a recorded signal; value and rate of the prescribed DOFs are the table's, the design gradient still flows
    (against central differences of the objective along a random direction)."""
    from difflexmm_amd import loading as L
    tt = np.linspace(0.0, TSIM, 11)          # (no output time falls on a knot: the rate there would be one-sided)
    sig = 2.0 * np.sin(np.pi * tt / tt[-1]) ** 2
    fw = P.QuadsFocusingForward(n1_blocks=N1, n2_blocks=N2, damping=damping(), loaded_side="right", input_shift=0, steps_per_interval=2 * SPI,
                                _lib=lib, **KW)
    fw.setup(excited_blocks_fn=L.Table(tt, sig))
    _, _, x = quads_forward(lib)
    sol = fw.solve(x)
    d = fw.driven_blocks_ids[0]
    assert np.allclose(sol.fields[:, 0, d, 0], np.interp(fw.timepoints, tt, sig), atol=1e-14)      # (the amplitude sign flip of the right side applies to the pulse only)
    slope = np.gradient(sig, tt)
    assert abs(sol.fields[1, 1, d, 0] - (np.interp(fw.timepoints[1] + 1e-9, tt, sig) - np.interp(fw.timepoints[1] - 1e-9, tt, sig)) / 2e-9) < 1e-4 * np.abs(slope).max()
    obj = P.TargetKineticEnergy(fw, (2, 2), (1, 0))
    v, g = obj.value_and_grad(x)
    rng = np.random.default_rng(1)
    dirn = tuple(rng.normal(size=a.shape) for a in x)
    eps = 1e-6
    vp = obj.value(tuple(a + eps * b for a, b in zip(x, dirn)))
    vm = obj.value(tuple(a - eps * b for a, b in zip(x, dirn)))
    fd = (vp - vm) / (2 * eps)
    an = sum((a * b).sum() for a, b in zip(g, dirn))
    assert v > 0 and abs(an - fd) < 1e-5 * abs(fd), (an, fd)
    with np.testing.assert_raises(TypeError):
        fw.setup(excited_blocks_fn=lambda t: 0.0)


def check_more_designs_than_batch(lib):
    """A list of designs longer than the engine's ``batch`` runs in consecutive engine calls (config 4 as written on one GPU: 64 designs whose
    smallest checkpoint does not fit at once): values and gradients equal those of one design at a time, in order."""
    fw2 = P.QuadsFocusingForward(n1_blocks=N1, n2_blocks=N2, damping=damping(), loaded_side="left", input_shift=0, steps_per_interval=SPI,
                                 batch=2, _lib=lib, **KW)
    fw1, _, x = quads_forward(lib)
    rng = np.random.default_rng(23)
    designs = [tuple(a + rng.uniform(-0.1, 0.1, a.shape) for a in x) for _ in range(4)]
    o2, o1 = P.TargetKineticEnergy(fw2, (2, 2), (1, 0)), P.TargetKineticEnergy(fw1, (2, 2), (1, 0))
    v, g = o2.value_and_grad(designs)
    assert v.shape == (4,) and len(g) == 4 and len(set(np.round(v / v.max(), 9))) == 4
    for m, d in enumerate(designs):
        v1, g1 = o1.value_and_grad(d)
        assert abs(v[m] - v1) <= 1e-13 * abs(v1)
        for a, b in zip(g[m], g1):
            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    # a length that is not a multiple of the batch is still refused by the solver
    with np.testing.assert_raises(Exception):
        o2.value_and_grad(designs[:3])
