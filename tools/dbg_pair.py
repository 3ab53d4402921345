import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.common import Case
from tests.test_gpu_pair_launches import _solve, FAST
c = Case("quads", 37, True, True, seed=21, cutoff_deg=42.0)
c.cp = c.cp._replace(constraint_params=FAST)
ts = np.linspace(0.0, 3e-4, 4)
mid = c.geo.n_blocks // 2
target = np.array([mid + 1, mid + 2], dtype=np.int32)
for env in ({"DFX_PAIR": "0"}, {"DFX_PAIR": "1", "DFX_CHECKPOINT": "records"}, {"DFX_PAIR": "1", "DFX_CHECKPOINT": "segments"}, {"DFX_PAIR": "f", "DFX_CHECKPOINT": "stages"}):
    out = _solve(c, ts, 7, target, env)
    print(env, "fwd", c.solver.stats["launches"], c.solver.stats.get("checkpoint_records"), "adj", out[3]["launches"], out[3].get("checkpoint_records"), out[1])
# checkpoint test
c = Case("quads", 16, True, True, seed=2, cutoff_deg=42.0)
c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
ts = np.linspace(0.0, 2e-4, 3)
y0 = np.zeros((2, 256, 3))
os.environ.pop("DFX_CHECKPOINT", None)
c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=10); print(c.solver.stats)
os.environ["DFX_TEST_FREE_BYTES"] = "1024"
c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=10); print(c.solver.stats)
c.solver(y0, np.linspace(0.0, 4e-4, 5), c.cp, keep_trajectory=True, steps_per_interval=10); print(c.solver.stats)
