"""BASELINE config 4 through the caller the reference uses (problems/kagome_focusing.py restated in difflexmm_amd/problems.py):
64x64-cell kagome, pulse on the left edge, target kinetic energy, B designs side by side on one GPU, forward + design gradient.
usage: python tools/c4_problem_timing.py [B] [steps]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difflexmm_amd import problems as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
n1 = n2 = 64
nb = 2 * n1 * n2
rho, ksh, kr, cell = 6.18e-9, 1.19, 1.5, 30.0
damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * 15.0 ** 2 * ksh)] * 2 + [2 * math.sqrt(0.02175026 * rho * 15.0 ** 4 * kr)]) * np.ones((nb, 1))
fw = P.KagomeFocusingForward(n1_cells=n1, n2_cells=n2, cell_size=cell, bond_length=2.25, k_stretch=120.0, k_shear=ksh, k_rot=kr,
                             density=rho, damping=damping, amplitude=7.5, loading_rate=30.0, input_delay=0.1 / 30, n_excited_blocks=2,
                             simulation_time=2.0 / 30, n_timepoints=41, use_contact=True, k_contact=1.5,
                             min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180, steps_per_interval=steps // 40, batch=B)
obj = P.TargetKineticEnergy(fw, (2, 2), (n1 // 6, n2 // 5))
rng = np.random.default_rng(0)
designs = [tuple(rng.uniform(-0.02 * cell, 0.02 * cell, s) for s in fw.geometry.design_shapes()) for _ in range(B)]
for rep in range(3):
    t0 = time.perf_counter()
    v, g = obj.value_and_grad(designs)
    wall = time.perf_counter() - t0
    sd = fw.solve_dynamics
    print(f"kagome {n1}x{n2} cells ({nb} units) x {B} designs, {steps // 40 * 40} steps: wall {wall:.2f} s, device fwd {sd.stats['kernel_ms']:.0f} ms + "
          f"adj {sd.adjoint_stats['kernel_ms']:.0f} ms, {steps // 40 * 40 * nb * B / wall:.3e} timesteps*units/s (host maps included), "
          f"device only {steps // 40 * 40 * nb * B / (1e-3 * (sd.stats['kernel_ms'] + sd.adjoint_stats['kernel_ms'])):.3e}; "
          f"checkpoint {sd.adjoint_stats.get('checkpoint_records')}/{sd.adjoint_stats.get('stage_checkpoint')}, streams {sd.adjoint_stats.get('streams')}, objective {np.atleast_1d(v)[0]:.3e}", flush=True)
