"""Would ONE engine that integrates the three inputs of config 5 side by side (members = inputs x designs) beat three engines of
`designs` members each?  Proxy without building it: three engines of M members (today's multi-input objective) against one engine of
3 M members of a single input -- the same number of solves per evaluation.
usage: python tools/merged_inputs_probe.py [M]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difflexmm_amd import problems as P

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def fw(side, shift, batch):
    f = P.QuadsFocusingForward(
        n1_blocks=24, n2_blocks=16, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9,
        damping=0.0186 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((384, 1)),
        amplitude=7.5, loading_rate=30.0, input_delay=0.1 / 30, n_excited_blocks=2, loaded_side=side, input_shift=shift,
        simulation_time=2.0 / 30, n_timepoints=41, use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180,
        cutoff_angle=-10 * math.pi / 180, steps_per_interval=100, batch=batch)
    f.setup()
    return f


rng = np.random.default_rng(0)
mi = P.MultiInputTargetKineticEnergy([fw(s, sh, M) for s, sh in (("left", 0), ("right", -2), ("bottom", -4))], (2, 2), (4, 3), weights=(1.0, 1.0, 1.0))
base = mi.forward.geometry.get_design_from_rotated_square(25 * math.pi / 180)
designs = [tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base) for _ in range(3 * M)]
for r in range(3):
    t0 = time.perf_counter()
    v, g = mi.value_and_grad(designs[:M])
    dev = sum(o.forward.solve_dynamics.stats["kernel_ms"] + o.forward.solve_dynamics.adjoint_stats["kernel_ms"] for o in mi.objectives)
    print(f"three engines x {M} members, round {r}: {time.perf_counter() - t0:.3f} s wall, device (sum of engines) {dev * 1e-3:.3f} s, "
          f"streams {mi.objectives[0].forward.solve_dynamics.adjoint_stats['streams']}", flush=True)
for o in mi.objectives:
    o.forward.solve_dynamics.engine.close()
one = fw("left", 0, 3 * M)
obj = P.TargetKineticEnergy(one, (2, 2), (4, 3))
for r in range(3):
    t0 = time.perf_counter()
    v, g = obj.value_and_grad(designs)
    sd = one.solve_dynamics
    print(f"one engine x {3 * M} members, round {r}: {time.perf_counter() - t0:.3f} s wall, device {1e-3 * (sd.stats['kernel_ms'] + sd.adjoint_stats['kernel_ms']):.3f} s, "
          f"streams {sd.adjoint_stats['streams']}", flush=True)
