"""Lifecycle stress on the HIP engine: many create/solve/destroy cycles, alternating adaptive and fixed-grid solves with
changing grids on one handle; free device memory must come back and results must repeat."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import Case

free0 = None
ref = None
for it in range(40):
    if it == 6:                      # after the runtime's own pools (code objects, kernarg and signal pools) have grown
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info(0)[0]
    c = Case("quads", 8, True, True, seed=3, lib=None, cutoff_deg=42.0, batch=1 + it % 3)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    y0 = c.random_state(0.05, 0.02, 5.0)
    s = c.solver
    for k in range(3):
        ts = np.linspace(0, 3e-4, 4 + k)
        a = s(y0, ts, cp)                                            # adaptive
        f = s(y0, ts, cp, keep_trajectory=True, steps_per_interval=5 + k)
        fb = np.ones_like(f)
        s.vjp(fb)
        g = s(y0, ts, cp, keep_trajectory=True)                      # frozen adaptive grid
        s.vjp(np.ones_like(g))
    probe = np.asarray(a)[0] if np.asarray(a).ndim == 5 else np.asarray(a)
    if it % 3 == 0:
        if ref is None:
            ref = probe.copy()
        assert np.array_equal(ref, probe), "results changed between cycles"
    s.engine.close()
    del c, s
free1 = torch.cuda.mem_get_info(0)[0]
print(f"free before {free0 >> 20} MiB, after {free1 >> 20} MiB, leaked {(free0 - free1) >> 20} MiB over 34 cycles")
assert free0 - free1 < 256 << 20
print("lifecycle ok")
