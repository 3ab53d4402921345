"""Config 4 at full size without any oracle: the design gradient of the target kinetic energy of two 64x64-cell kagome designs (contact, damping, pulse)
over K fixed Dopri5 steps through the problem layer (reverse sweep + host-side design maps) against central finite differences of the same
objective along a random direction of the three shift fields.   usage: python tools/fd_check_c4.py [K] [eps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
fw, obj, K = bench.c4_problem(2, K)
designs = []
for seed in (100, 101):
    rng = np.random.default_rng(seed)
    designs.append(tuple(rng.uniform(-0.3, 0.3, sh) for sh in fw.geometry.design_shapes()))
rng = np.random.default_rng(7)
direction = [tuple(rng.normal(size=a.shape) for a in d) for d in designs]
t0 = time.perf_counter()
v, g = obj.value_and_grad(designs)
sd = fw.solve_dynamics
an = np.array([sum(float((a * b).sum()) for a, b in zip(gm, dm)) for gm, dm in zip(g, direction)])
plus = [tuple(a + eps * d for a, d in zip(dm, dd)) for dm, dd in zip(designs, direction)]
minus = [tuple(a - eps * d for a, d in zip(dm, dd)) for dm, dd in zip(designs, direction)]
fd = (np.asarray(obj.value(plus)) - np.asarray(obj.value(minus))) / (2 * eps)
lvl = {1: "records", 2: "segments"}.get(sd.adjoint_stats.get("checkpoint_records", 0)) or ("stages" if sd.adjoint_stats.get("stage_checkpoint") else "state")
print(f"K = {K} steps, checkpoint {lvl}, kernels {bench.BUILD_NAMES.get(int(sd.adjoint_stats.get('tile_kernels', 0)))}, objective {np.asarray(v)}, {time.perf_counter() - t0:.1f} s")
for m in range(2):
    print(f"  design {m}: adjoint {an[m]:+.12e}   finite differences {fd[m]:+.12e}   relative difference {abs(an[m] - fd[m]) / abs(fd[m]):.2e}")
