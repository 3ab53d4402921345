"""-m gpu: dfx_kinetic_value_and_grad_device (gradients left in HBM, device pointers handed out) against dfx_kinetic_value_and_grad
(the same gradients as pinned host views): bitwise, quads (accumulator layout as is) and kagome (3 of 4 node slots: packed on the
device), and the entries the library assembles on the host are refused."""
import numpy as np
import pytest

from .common import Case
from .test_gpu_pair_launches import FAST

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lattice,n,batch", [("quads", 12, 3), ("kagome", 7, 2)])
def test_device_resident_gradients_equal_host_views(hip_lib, lattice, n, batch):
    c = Case(lattice, n, True, True, seed=5, cutoff_deg=42.0 if lattice == "quads" else 125.0, batch=batch)
    cps = [c.cp._replace(constraint_params=dict(FAST, amplitude=7.5 * (1 + 0.05 * m))) for m in range(batch)]
    ts = np.linspace(0.0, 3e-4, 4)
    c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, cps, keep_trajectory=True, steps_per_interval=7)
    eng = c.solver.engine
    mid = c.geo.n_blocks // 2
    target = np.array([mid + 1, mid + 2], dtype=np.int32)
    which = ("centroid_node_vectors", "void_angle0", "inertia", "damping")
    obj_h, g_h, _ = eng.kinetic_value_and_grad(target, which=which)
    g_h = {k: np.array(v) for k, v in g_h.items()}
    obj_d, g_d, st = eng.kinetic_value_and_grad(target, which=which, device=True)
    assert np.array_equal(obj_h, obj_d) and st["launches"] > 0
    for k in which:
        a = g_d[k].to_host()
        assert a.shape == g_h[k].shape and np.array_equal(a, g_h[k]), k
    assert np.abs(g_h["centroid_node_vectors"]).max() > 0 and np.abs(g_h["void_angle0"]).max() > 0
    with pytest.raises(RuntimeError, match="assembled on the host"):
        eng.kinetic_value_and_grad(target, which=("k_bond",), device=True)
