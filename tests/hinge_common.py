"""problems/hinge_characterization.py on the engine (difflexmm_amd/hinge.py) against the oracle twin: force-displacement curves of the
three tests, the squared-error objective and its gradient w.r.t. (k_stretch, k_shear, k_rot) by autograd through the unrolled oracle
-- the dynamics, the elastic reaction force and both of its derivatives."""
import math

import numpy as np
import torch

from difflexmm_amd import hinge as H
from oracle import ref_problems as RP

SPI, NT = 10, 5
KW = dict(spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9, damping=0.2, amplitude=1.2,
          loading_rate=1500.0, n_timepoints=NT, use_contact=True, k_contact=1.5, min_angle=5 * math.pi / 180, cutoff_angle=48 * math.pi / 180)
K = (100.0, 1.5, 1.2)


def forwards(lib):
    fws = [H.HingeForward(n1_cells=2, n2_cells=2, initial_angle=25 * math.pi / 180, loading_type=lt, steps_per_interval=SPI,
                          force_multiplier=-1.0 if lt == "compression" else 1.0, _lib=lib, **KW) for lt in ("tension", "compression", "shear")]
    ofs = [RP.HingeForward("rotated_squares", 2, 2, KW["spacing"], KW["bond_length"], (25 * math.pi / 180,), KW["k_stretch"], KW["density"],
                           KW["damping"], lt, KW["amplitude"], KW["loading_rate"], NT, force_multiplier=-1.0 if lt == "compression" else 1.0,
                           use_contact=True, k_contact=1.5, min_angle=KW["min_angle"], cutoff_angle=KW["cutoff_angle"])
           for lt in ("tension", "compression", "shear")]
    return fws, ofs


def check_force_displacement_and_fit_gradient(lib, tol=1e-9):
    fws, ofs = forwards(lib)
    rng = np.random.default_rng(4)
    targets = {}
    for fw in fws:
        fw.setup()
        u = np.linspace(0, KW["amplitude"], 9) * (-1.0 if fw.loading_type == "compression" else 1.0)
        targets[fw.loading_type] = np.array([u, 3.0 * np.abs(u) + rng.uniform(-0.2, 0.2, 9), 0.1 * np.ones(9)])
    opt = H.HingeResponseError(fws, targets)
    v, g = opt.value_and_grad(K)
    kt = [torch.tensor(k, dtype=torch.float64, requires_grad=True) for k in K]
    for fw, of in zip(fws, ofs):
        assert np.array_equal(fw.constrained_block_DOF_pairs, of.constrained_block_DOF_pairs)
        assert np.array_equal(fw.reaction_block_DOF_pairs, of.reaction_block_DOF_pairs)
        sol, cp = fw.solve(K)
        got = fw.force_displacement(sol, cp)
        au, ff = of.force_displacement(kt, SPI)
        assert np.abs(got[0] - au.detach().numpy()).max() < 1e-14
        ref = ff.detach().numpy()
        assert np.abs(ref).max() > 1e-3 and np.abs(got[1] - ref).max() < tol * np.abs(ref).max(), (fw.loading_type, got[1], ref)
    ov = RP.hinge_response_squared_error(ofs, opt.target_forces, kt, SPI)
    og = torch.autograd.grad(ov, kt)
    assert abs(v - ov.item()) < tol * abs(ov.item()) and abs(opt.objective_fn(K) - v) < 1e-12 * abs(v)
    for a, b in zip(g, og):
        assert abs(a - b.item()) < tol * max(abs(x.item()) for x in og), (g, [x.item() for x in og])
    return opt


def check_fit_loops(lib):
    """Both loops of the reference decrease the error; the bookkeeping matches (objective_values, design_values, fitted_responses)."""
    fws, _ = forwards(lib)
    for fw in fws:
        fw.setup()
    # targets = the responses of a sample with known stiffnesses: the fit must move towards them
    truth = (110.0, 1.3, 1.35)
    targets = {fw.loading_type: np.vstack([fw.force_displacement(*fw.solve(truth)), np.ones(NT)]) for fw in fws}
    opt = H.HingeResponseError(fws, targets)
    opt.run_optimization_nlopt(K, 6, lower_bound=[50.0, 0.5, 0.5], upper_bound=[200.0, 3.0, 3.0])
    assert len(opt.objective_values) <= 6 and min(opt.objective_values) < 0.5 * opt.objective_values[0]
    assert set(opt.fitted_responses) == {"tension", "compression", "shear"} and opt.fitted_responses["shear"].shape == (2, NT)
    gd = H.HingeResponseError(fws, targets)
    g0 = np.abs(gd.value_and_grad(K)[1]).max()
    gd.run_optimization_GD(K, 3, step_size=0.05 / g0, lower_bound=0.5, upper_bound=200.0)
    assert len(gd.objective_values) == 3 and len(gd.design_values) == 4 and gd.objective_values[-1] < gd.objective_values[0]
    d = H.HingeResponseError.from_dict(opt.to_dict(), _lib=lib)
    assert len(d.forward_problems) == 3 and d.objective_values == opt.objective_values


def check_quads_sample(lib, tol=1e-9):
    """ForwardProblemQuads (:281-545): a random quad sample in shear -- response and fit gradient against the oracle twin."""
    from difflexmm_amd.geometry import QuadGeometry
    n1, n2 = 4, 4
    g = QuadGeometry(n1, n2, KW["spacing"], KW["bond_length"])
    rng = np.random.default_rng(8)
    hs, vs = (b + rng.uniform(-0.3, 0.3, b.shape) for b in g.get_design_from_rotated_square(25 * math.pi / 180))
    fw = H.HingeQuadsForward(n1_blocks=n1, n2_blocks=n2, horizontal_shifts=hs, vertical_shifts=vs, loading_type="shear",
                             steps_per_interval=SPI, _lib=lib, **KW)
    of = RP.HingeForward("quads", n1, n2, KW["spacing"], KW["bond_length"], (hs, vs), KW["k_stretch"], KW["density"], KW["damping"], "shear",
                         KW["amplitude"], KW["loading_rate"], NT, use_contact=True, k_contact=1.5, min_angle=KW["min_angle"],
                         cutoff_angle=KW["cutoff_angle"])
    u = np.linspace(0, KW["amplitude"], 7)
    opt = H.HingeResponseError([fw], {"shear": np.array([u, 2.0 * u, np.ones(7)])})
    v, grad = opt.value_and_grad(K)
    kt = [torch.tensor(k, dtype=torch.float64, requires_grad=True) for k in K]
    ov = RP.hinge_response_squared_error([of], opt.target_forces, kt, SPI)
    og = torch.autograd.grad(ov, kt)
    assert abs(v - ov.item()) < tol * abs(ov.item())
    for a, b in zip(grad, og):
        assert abs(a - b.item()) < tol * max(abs(x.item()) for x in og)
