"""Where the wall time of one bench execute() goes when K is small (fixed per-solve overheads)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ.setdefault("DFX_STREAMS", "2")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fw, obj, designs = bench.c3_problem(128, 3, 16)
eng = fw.solve_dynamics.engine
eng.reserve(max(K, 250), 4, keep_trajectory=True)
bench.run_once(fw, obj, designs, 250)
bench.run_once(fw, obj, designs, K)
bench.prepare(fw, designs, K)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, st_f = eng.forward(np.zeros((eng.batch, 2, eng.n_blocks, 3)), fw.timepoints, fw.step_counts, keep_trajectory=True, want_fields=False)
    t1 = time.perf_counter()
    o = eng.objective_kinetic(obj.target_blocks)
    t2 = time.perf_counter()
    grads, st_a = eng.adjoint_kinetic(obj.target_blocks, which=("centroid_node_vectors", "void_angle0", "inertia"))
    t3 = time.perf_counter()
    print(f"K={K}: forward {1e3*(t1-t0):.1f} ms (device {st_f['kernel_ms']:.1f}), objective {1e3*(t2-t1):.1f} ms, adjoint {1e3*(t3-t2):.1f} ms (device {st_a['kernel_ms']:.1f})")
