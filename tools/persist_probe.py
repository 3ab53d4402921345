"""Persistent stage loop (dfx_persist.h) against one launch per stage, on the device's own clock: forward and reverse time per stage for
a lattice / ensemble width / checkpoint level.   python tools/persist_probe.py LATTICE N BATCH STEPS [contact=1] [adjoint=1] [VAR=VAL ...]
Each remaining argument is an environment setting of one more arm (e.g. DFX_PERSIST_CHUNKS=4); the two standard arms are DFX_PERSIST=0 / 1."""
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from common import Case  # noqa: E402

lattice, n, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
contact = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
adjoint = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
arms = [{"DFX_PERSIST": "0"}, {"DFX_PERSIST": "1"}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[7:]]
T = 5
ts = np.linspace(0.0, 1e-3, T)
spi = max(1, steps // (T - 1))
for env in arms:
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = Case(lattice, n, True, contact, seed=100, lib=None, cutoff_deg=125.0 if lattice == "kagome" else -10.0, batch=B)
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=1000.0, input_delay=1e-5))
        nb = c.geo.n_blocks
        mid = nb // 2
        target = np.array([mid + 1, mid + 2], dtype=np.int32)
        y0 = np.zeros((2, nb, 3))
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            c.solver(y0, ts, [cp] * B if B > 1 else cp, keep_trajectory=adjoint, steps_per_interval=spi, want_fields=False)
            st = dict(c.solver.stats)
            sa = {}
            if adjoint:
                obj, raw = c.solver.kinetic_energy_value_and_raw(target)
                sa = dict(c.solver.adjoint_stats)
            wall = time.perf_counter() - t0
            row = (st["kernel_ms"], sa.get("kernel_ms", 0.0), wall, st, sa)
            if best is None or row[0] + row[1] < best[0] + best[1]:
                best = row
        f_ms, a_ms, wall, st, sa = best
        n_stage = spi * (T - 1) * 6
        units = spi * (T - 1) * nb * B
        print(f"{lattice} {n} x {B} members, {spi * (T - 1)} steps, {env}: fwd {1e3 * f_ms / n_stage:.2f} us/stage (build {st['tile_kernels']}, "
              f"{st['launches']} launches), adj {1e3 * a_ms / n_stage:.2f} us/stage (build {sa.get('tile_kernels')}, level records={sa.get('checkpoint_records')}), "
              f"device fwd {units / (f_ms * 1e-3):.3e}" + (f", fwd+adj {units / ((f_ms + a_ms) * 1e-3):.3e}" if adjoint else "") + f" units*steps/s, wall {wall * 1e3:.1f} ms", flush=True)
        del c
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
