"""Shared body of the quads_spin parity tests (problems/quads_spin.py:210-222, 391-428; difflexmm/energy.py:502-519): harmonic
drive + target angular momentum, value and design gradient against autograd through the unrolled oracle."""
import math

import numpy as np
import torch

from difflexmm_amd import problems as P
from oracle import ref_problems as RP

N1, N2, SPI, NT, TSIM = 6, 5, 10, 4, 9e-4
KW = dict(spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9, amplitude=4.0, loading_rate=1500.0,
          input_delay=5e-5, n_excited_blocks=1, simulation_time=TSIM, n_timepoints=NT, use_contact=True, k_contact=1.5,
          min_angle=5 * math.pi / 180, cutoff_angle=45 * math.pi / 180)


def damping():
    return 0.05 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((N1 * N2, 1))


def check_angular_momentum_value_and_gradient(lib, tol=1e-9):
    fw = P.QuadsSpinForward(n1_blocks=N1, n2_blocks=N2, damping=damping(), loaded_side="left", input_shift=0, steps_per_interval=SPI,
                            _lib=lib, **KW)
    fw.setup()
    rng = np.random.default_rng(9)
    base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
    x = tuple(b + rng.uniform(-0.2, 0.2, b.shape) for b in base)
    obj = P.TargetAngularMomentum(fw, (2, 2), (1, 0), spin_center="center", reference_design=x)
    v, g = obj.value_and_grad(x)
    ofw = RP.ForwardProblem("quads", N1, N2, KW["spacing"], KW["bond_length"], KW["k_stretch"], KW["k_shear"], KW["k_rot"], KW["density"],
                            damping(), KW["amplitude"], KW["loading_rate"], KW["input_delay"], 1, TSIM, NT, "left", 0, use_contact=True,
                            k_contact=1.5, min_angle=KW["min_angle"], cutoff_angle=KW["cutoff_angle"], signal=RP.harmonic_signal)
    tb = RP.quads_target_blocks(N1, N2, (2, 2), (1, 0))
    assert np.array_equal(tb, obj.target_blocks)
    center = ofw.geometry.block_centroids(*[torch.tensor(a) for a in x])[torch.as_tensor(tb)].mean(0).numpy()       # quads_spin.py:400-402
    assert np.allclose(center, obj.spin_center, rtol=1e-14)
    xt = [torch.tensor(a, requires_grad=True) for a in x]
    ov = RP.target_angular_momentum(ofw, xt, tb, center, SPI)
    og = torch.autograd.grad(ov, xt)
    assert abs(ov.item()) > 0 and abs(v - ov.item()) < tol * abs(ov.item()), (v, ov.item())
    for a, b in zip(g, og):
        assert np.abs(a - b.numpy()).max() < tol * np.abs(b.numpy()).max()
    # the drive is the harmonic signal: still on after one period (a pulse would have stopped)
    sol = fw.solve(x)
    t_late = fw.timepoints[-1]
    assert t_late - KW["input_delay"] > 1.0 / KW["loading_rate"]
    expect = KW["amplitude"] * 0.5 * (1 - math.cos(2 * math.pi * KW["loading_rate"] * (t_late - KW["input_delay"])))
    assert abs(sol.fields[-1, 0, fw.driven_blocks_ids[0], 0] - expect) < 1e-12 * abs(expect)
