"""Full-size gradient check that needs no oracle: the design gradient of the target kinetic energy of C3 (128x128 quads, contact, damping,
pulse drive) over K fixed Dopri5 steps from the reverse sweep, against central finite differences of the same objective along a random
design direction.   usage: python tools/fd_check_fullsize.py [K] [eps]      (DFX_CHECKPOINT selects the level of the reverse sweep)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from difflexmm_amd.problems import design_gradients

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
fw, obj, designs = bench.c3_problem(128, 3, 2)
rng = np.random.default_rng(7)
direction = [tuple(rng.normal(size=a.shape) for a in d) for d in designs]


def value(ds, adjoint):
    bench.prepare(fw, ds, K)
    res = bench.execute(fw, obj, adjoint=adjoint)
    return res


t0 = time.perf_counter()
res = value(designs, True)
raw = {k: np.array(v) for k, v in res["grads"].items()}
grads = design_gradients(fw, designs, raw)
an = np.array([sum(float((g * d).sum()) for g, d in zip(gm, dm)) for gm, dm in zip(grads, direction)])


def objective_only(ds):
    bench.prepare(fw, ds, K)
    eng = fw.solve_dynamics.engine
    eng.forward(None, fw.timepoints, fw.step_counts, keep_trajectory=True, want_fields=False)
    o, _, _ = eng.kinetic_value_and_grad(obj.target_blocks, which=("inertia",))
    return np.array(o)


plus = [tuple(a + eps * d for a, d in zip(dm, dd)) for dm, dd in zip(designs, direction)]
minus = [tuple(a - eps * d for a, d in zip(dm, dd)) for dm, dd in zip(designs, direction)]
fd = (objective_only(plus) - objective_only(minus)) / (2 * eps)
print(f"K = {K} steps, checkpoint {res.get('checkpoint')}, objective {np.array(res['objective'])}, {time.perf_counter() - t0:.1f} s")
for m in range(2):
    print(f"  member {m}: adjoint {an[m]:+.12e}   finite differences {fd[m]:+.12e}   relative difference {abs(an[m] - fd[m]) / abs(fd[m]):.2e}")
