"""Where a short solve spends its time: the driver's bench call (`--steps 20`) repeated, host wall per engine call against the
device time each reports.  usage: python tools/k20_probe.py [K] [members] [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
os.environ.setdefault("DFX_STREAMS", "2")
fw, obj, des = bench.c3_problem(128, 3, M)
eng = fw.solve_dynamics.engine
eng.reserve(max(K, 250), max(K, 250) // bench.SPI + 2, keep_trajectory=True)
bench.run_once(fw, obj, des, K)
bench.prepare(fw, des, K)
state0 = np.zeros((eng.batch, 2, eng.n_blocks, 3))
for r in range(R):
    bench.spin_up(fw)
    t0 = time.perf_counter()
    _, sf = eng.forward(None, fw.timepoints, fw.step_counts, keep_trajectory=True, want_fields=False)
    t1 = time.perf_counter()
    t2 = t1
    o, g, sa = eng.kinetic_value_and_grad(obj.target_blocks, which=("centroid_node_vectors", "void_angle0", "inertia"))
    t3 = time.perf_counter()
    print(f"K={K} M={M}: forward wall {1e3*(t1-t0):.2f} ms (device {sf['kernel_ms']:.2f}), objective {1e3*(t2-t1):.2f} ms, "
          f"adjoint wall {1e3*(t3-t2):.2f} ms (device {sa['kernel_ms']:.2f}), total {1e3*(t3-t0):.2f} ms; "
          f"value {K*16384*M/(t3-t0):.3e}", flush=True)
