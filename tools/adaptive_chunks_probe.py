"""Adaptive solve that keeps its accepted steps, members in several launches of the adaptive loop, against the stage-launch controller.
   python tools/adaptive_chunks_probe.py N BATCH [only=1|0]"""
import faulthandler, os, sys, numpy as np
faulthandler.enable()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from common import Case
FAST = dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5)
def run(persist, n, batch):
    os.environ["DFX_PERSIST"] = persist
    c = Case("quads", n, True, True, seed=17, cutoff_deg=42.0, batch=batch)
    cps = [c.cp._replace(constraint_params=dict(FAST, amplitude=7.5 / (1 + 0.6 * m))) for m in range(batch)]
    ts = np.linspace(0.0, 3e-4, 21)
    print("solve", persist, flush=True)
    f = np.array(c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, cps if batch > 1 else cps[0], keep_trajectory=True))
    st = dict(c.solver.stats)
    print("forward done", {k: st.get(k) for k in ("tile_kernels", "launches", "steps", "step_control")}, flush=True)
    mid = c.geo.n_blocks // 2
    obj, raw = c.solver.kinetic_energy_value_and_raw(np.array([mid + 1, mid + 2], dtype=np.int32))
    sa = dict(c.solver.adjoint_stats)
    print("reverse done", sa.get("tile_kernels"), sa.get("launches"), flush=True)
    out = (f, st, np.atleast_1d(obj).copy(), {k: np.array(v) for k, v in raw.items()}, sa)      # (raw: views of the engine's pinned result area)
    c.solver.engine.close()
    return out
n, batch = int(sys.argv[1]), int(sys.argv[2])
only = sys.argv[3] if len(sys.argv) > 3 else None
if only is not None:
    run(only, n, batch); sys.exit(0)
a = run("1", n, batch); b = run("0", n, batch)
rel = lambda x, y: float(np.abs(x - y).max() / max(1e-300, np.abs(y).max()))
print("fields rel", rel(a[0], b[0]), "objective rel", rel(a[2], b[2]), "grads", {k: rel(a[3][k], b[3][k]) for k in a[3]})
