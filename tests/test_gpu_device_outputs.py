"""-m gpu: dfx_kinetic_value_and_grad_device (gradients left in HBM, device pointers handed out) against dfx_kinetic_value_and_grad
(the same gradients as pinned host views): bitwise, quads (accumulator layout as is) and kagome (3 of 4 node slots: packed on the
device), and the entries the library assembles on the host are refused."""
import numpy as np
import pytest

from .common import Case
FAST = dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5)

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


@pytest.mark.parametrize("device", [False, True])
def test_fused_forward_value_and_grad_equals_the_two_calls(hip_lib, device):
    """dfx_forward_kinetic_value_and_grad (the host does not wait for the forward pass before it enqueues the reverse sweep) against
    dfx_forward_grid + dfx_kinetic_value_and_grad: bit for bit, statistics included; ragged steps per interval."""
    c = Case("quads", 12, True, True, seed=5, cutoff_deg=42.0, batch=3)
    cps = [c.cp._replace(constraint_params=dict(FAST, amplitude=7.5 * (1 + 0.05 * m))) for m in range(3)]
    ts = np.linspace(0.0, 3e-4, 4)
    spis = np.array([7, 5, 9], dtype=np.int32)
    c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, cps, keep_trajectory=True, steps_per_interval=7)      # packs and uploads the parameters
    eng = c.solver.engine
    mid = c.geo.n_blocks // 2
    target = np.array([mid + 1, mid + 2], dtype=np.int32)
    which = ("centroid_node_vectors", "void_angle0", "inertia")
    _, st_f = eng.forward(None, ts, spis, keep_trajectory=True, want_fields=False)
    obj2, g2, st_a = eng.kinetic_value_and_grad(target, which=which)
    g2 = {k: np.array(v) for k, v in g2.items()}
    obj1, g1, sf, sa = eng.forward_kinetic_value_and_grad(None, ts, spis, target, which=which, device=device)
    assert np.array_equal(obj1, obj2) and obj1.min() > 0
    for k in which:
        a = g1[k].to_host() if device else np.array(g1[k])
        assert np.array_equal(a, g2[k]), k
    assert sf["steps"] == st_f["steps"] == 21 and sf["launches"] == st_f["launches"] and sa["launches"] == st_a["launches"]
    assert sf["kernel_ms"] > 0 and sa["kernel_ms"] > 0


def test_fused_call_reports_a_blown_up_forward_pass(hip_lib):
    """The fused call checks the forward pass's non-finite flag at its end (recipe of test_unstable_step_reports_an_error)."""
    c = Case("quads", 6, True, False, seed=1, lib=None)
    c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=300.0, input_delay=0.0))
    ts = np.linspace(0.0, 3e-4, 3)
    c.solver(np.zeros((2, 36, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=7)      # packs and uploads the parameters
    eng = c.solver.engine
    target = np.array([1, 2], dtype=np.int32)
    with pytest.raises(RuntimeError, match="non-finite"):
        eng.forward_kinetic_value_and_grad(None, np.linspace(0, 0.5, 3), 20, target, which=("inertia",))      # h = 12.5 ms >> 1/omega_max
    obj, g, sf, sa = eng.forward_kinetic_value_and_grad(None, ts, 7, target, which=("inertia",))             # the handle still works
    assert np.isfinite(obj).all() and np.isfinite(g["inertia"]).all() and sf["steps"] == 14


def test_many_target_blocks_take_the_copy_path(hip_lib, cpu_lib):
    """The reverse sweep's prelude kernel takes up to 32 target blocks by value; more are uploaded with a copy.  40 targets on 12 x 12
    quads, HIP against the CPU port (objective and every gradient), and against the same call with 8 targets to make sure the two
    objectives differ (the extra blocks are really read)."""
    ts = np.linspace(0.0, 3e-4, 4)
    res = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 12, True, True, seed=5, lib=lib, cutoff_deg=42.0)
        c.cp = c.cp._replace(constraint_params=FAST)
        c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=7)
        targets = np.arange(20, 60, dtype=np.int32)
        obj, g, _ = c.solver.engine.kinetic_value_and_grad(targets, which=("centroid_node_vectors", "void_angle0", "inertia"))
        res[name] = (float(np.atleast_1d(obj)[0]), {k: np.array(v) for k, v in g.items()})
        if lib is None:
            obj8, _, _ = c.solver.engine.kinetic_value_and_grad(targets[:8], which=("inertia",))
            assert float(np.atleast_1d(obj8)[0]) < res[name][0]
    assert abs(res["hip"][0] - res["cpu"][0]) < 1e-10 * abs(res["cpu"][0]) and res["cpu"][0] > 0
    for k in res["cpu"][1]:
        d = np.abs(res["hip"][1][k] - res["cpu"][1][k]).max()
        assert d <= 1e-9 * np.abs(res["cpu"][1][k]).max(), (k, d)
