"""The reference's default call (adaptive Dormand-Prince, odeint semantics) at C3 size: attempted steps per second against the fixed
grid.  usage: python tools/adaptive_probe.py [members] [n_intervals] [rtol]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 2
RTOL = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
fw, obj, designs = bench.c3_problem(128, 3, M)
bench.prepare(fw, designs, NI * bench.SPI)
eng = fw.solve_dynamics.engine
ts = np.asarray(fw.timepoints)
n_units = 128 * 128
state0 = np.zeros((M, 2, n_units, 3))
for rep in range(3):
    t0 = time.perf_counter()
    f, st = eng.forward_adaptive(state0, ts, RTOL, RTOL)
    wall = time.perf_counter() - t0
    counts = eng.adaptive_step_counts()
    att = st.get("attempts", st.get("n_steps", 0))
    print(f"adaptive rtol={RTOL:g}: wall {wall * 1e3:.1f} ms, device {st['kernel_ms']:.1f} ms, launches {st['launches']}, stats {dict((k, st[k]) for k in st if k not in ('kernel_ms',))}")
    print(f"   accepted steps per member and interval: min {counts.min()} max {counts.max()}; "
          f"device time per launch {1e3 * st['kernel_ms'] / max(1, st['launches']):.2f} us", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    f2, st2 = eng.forward(None, ts, bench.SPI, keep_trajectory=False, want_fields=True)
    wall = time.perf_counter() - t0
    print(f"fixed grid {bench.SPI} steps/interval: wall {wall * 1e3:.1f} ms, device {st2['kernel_ms']:.1f} ms, launches {st2['launches']}, "
          f"per step {1e3 * st2['kernel_ms'] / (NI * bench.SPI):.1f} us", flush=True)
