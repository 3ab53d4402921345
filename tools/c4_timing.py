"""BASELINE config 4 on one GPU: 64x64-cell kagome (8 192 triangles), contact + damping, pulse drive, 8 designs side by side,
forward + gradient of the target kinetic energy w.r.t. the three shift fields; prints timesteps*units/s."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from common import Case

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
c = Case("kagome", 64, True, True, seed=100, lib=None, cutoff_deg=125.0, batch=B)
cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=30.0, input_delay=0.1 / 30))
ts = np.linspace(0.0, 3.0 / 30, 41)
spi = steps // 40
mid = 2 * 64 * 32
target = np.array([mid + 2, mid + 3, mid + 4, mid + 5], dtype=np.int32)
y0 = np.zeros((2, c.geo.n_blocks, 3))
n = c.geo.n_blocks
for which in ("every ControlParams leaf", "design subset"):
    for rep in range(2):
        t0 = time.perf_counter()
        c.solver(y0, ts, [cp] * B, keep_trajectory=True, steps_per_interval=spi)
        if which == "design subset":      # what a design optimisation needs: node vectors, undeformed void angles, inertia
            obj, raw = c.solver.kinetic_energy_value_and_raw(target)
        else:
            obj, trees, _ = c.solver.kinetic_energy_value_and_vjp(target)
        wall = time.perf_counter() - t0
    st, sa = c.solver.stats, c.solver.adjoint_stats
    print(f"gradient w.r.t. {which}: wall {wall:.2f} s, device fwd {st['kernel_ms']:.0f} ms + adj {sa['kernel_ms']:.0f} ms, "
          f"{spi * 40 * n * B / wall:.3e} timesteps*units/s")
print({k: sa.get(k) for k in ("launches", "streams", "stage_checkpoint", "checkpoint_records")}, {k: st.get(k) for k in ("launches", "streams")})
print(f"kagome 64x64 cells ({n} units) x {B} designs, {spi * 40} steps: wall {wall:.2f} s, device fwd {st['kernel_ms']:.0f} ms + adj {sa['kernel_ms']:.0f} ms, "
      f"{spi * 40 * n * B / wall:.3e} timesteps*units/s (fwd+grad, host included), fwd only {spi * 40 * n * B / (st['kernel_ms'] * 1e-3):.3e}, objective {np.atleast_1d(obj)[0]:.3e}")
