import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import Case
mode = sys.argv[1]
def cycle(it):
    c = Case("quads", 8, True, True, seed=3, lib=None, cutoff_deg=42.0, batch=2)
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    y0 = c.random_state(0.05, 0.02, 5.0); s = c.solver; ts = np.linspace(0, 3e-4, 5)
    if mode in ("create",): pass
    if mode in ("adaptive", "all"): s(y0, ts, cp)
    if mode in ("fixed", "all", "fixed_adj"): f = s(y0, ts, cp, keep_trajectory=True, steps_per_interval=5)
    if mode in ("fixed_adj", "all"): s.vjp(np.ones_like(f))
    s.engine.close()
cycle(0); cycle(1)
torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info(0)[0]
for it in range(30): cycle(it)
torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info(0)[0]
print(mode, os.environ.get("DFX_NO_GRAPH"), "leak per cycle KiB:", (f0 - f1) / 30 / 1024)
