"""Where does the wall time of one engine call go for config 5's lattice (24x16 quads, M members, 4 000 steps)?  wall of forward() and of
kinetic_value_and_grad() against the device time they report.   usage: python tools/c5_call_probe.py [M]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difflexmm_amd import problems as P
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = P.QuadsFocusingForward(
    n1_blocks=24, n2_blocks=16, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9,
    damping=0.0186 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((384, 1)),
    amplitude=7.5, loading_rate=30.0, input_delay=0.1 / 30, n_excited_blocks=2, loaded_side="left", input_shift=0,
    simulation_time=2.0 / 30, n_timepoints=41, use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180,
    cutoff_angle=-10 * math.pi / 180, steps_per_interval=100, batch=M)
f.setup()
obj = P.TargetKineticEnergy(f, (2, 2), (4, 3))
rng = np.random.default_rng(0)
base = f.geometry.get_design_from_rotated_square(25 * math.pi / 180)
designs = [tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base) for _ in range(M)]
sd = f.solve_dynamics
cps = [f.control_params(d) for d in designs]
flats = [sd._flatten(cp) for cp in cps]
eng = sd.engine
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_params(**{k: np.stack([fl[k] for fl in flats]) for k in flats[0]})
    t1 = time.perf_counter()
    _, st = eng.forward(None, f.timepoints, 100, keep_trajectory=True, want_fields=False)
    t2 = time.perf_counter()
    o, g, sa = eng.kinetic_value_and_grad(obj.target_blocks, which=("centroid_node_vectors", "void_angle0", "inertia"))
    t3 = time.perf_counter()
    print(f"M = {M}: set_params {1e3 * (t1 - t0):.0f} ms; forward wall {1e3 * (t2 - t1):.0f} ms (device {st['kernel_ms']:.0f} ms, {st['launches']} launches, streams {st['streams']}, "
          f"records {st['checkpoint_records']} stages {st['stage_checkpoint']}); reverse wall {1e3 * (t3 - t2):.0f} ms (device {sa['kernel_ms']:.0f} ms, {sa['launches']} launches)", flush=True)
