"""-m gpu: the HIP engine (through the C ABI, libdfx.so) against the oracle on identical seeded inputs."""
import numpy as np
import pytest

from . import parity
from .common import Case, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lattice,n", [("quads", 4), ("quads", 9), ("kagome", 3), ("kagome", 5)])
@pytest.mark.parametrize("nonlinear", [True, False])
@pytest.mark.parametrize("contact", [False, True])
def test_rhs_and_vjp_match_autograd(hip_lib, lattice, n, nonlinear, contact):
    parity.check_rhs_and_vjp(None, lattice, n, nonlinear, contact)


@pytest.mark.parametrize("lattice,n,integrator", [("quads", 4, "dopri5"), ("kagome", 3, "rk4"), ("quads", 3, "rk4"),
                                                  ("kagome", 4, "dopri5")])
def test_trajectory_and_discrete_adjoint(hip_lib, lattice, n, integrator):
    parity.check_trajectory_and_adjoint(None, lattice, n, integrator)


def test_linearized_no_contact_adjoint(hip_lib):
    parity.check_trajectory_and_adjoint(None, "quads", 4, "dopri5", nonlinear=False, contact=False)


def _solve(c, y0, ts, spi, target=None, keep=True):
    fields = c.solver(y0, ts, c.cp, keep_trajectory=keep, steps_per_interval=spi)
    return fields


def test_hip_matches_cpu_port_32x32_with_segments(hip_lib, cpu_lib):
    """32x32 quads (config C2 physics + contact), 600 steps per interval so that one interval spans several
    hipGraph segments; HIP vs the CPU port of the oracle, forward fields and kinetic-energy gradient."""
    ts = np.linspace(0.0, 2.4e-3, 3)
    res = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 32, True, True, seed=2, lib=lib)
        c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=800.0, input_delay=1e-5))
        f = c.solver(np.zeros((2, 1024, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=600)
        obj, tree, s0 = c.solver.kinetic_energy_value_and_vjp(np.array([16 * 32 + 2, 16 * 32 + 3]))
        res[name] = (f, obj, tree.geometrical_params.centroid_node_vectors, tree.constraint_params["amplitude"])
    assert np.abs(res["cpu"][0]).max() > 1e-3
    assert relerr(res["hip"][0], res["cpu"][0]) < 1e-9
    assert abs(res["hip"][1] - res["cpu"][1]) / abs(res["cpu"][1]) < 1e-9
    assert relerr(res["hip"][2], res["cpu"][2]) < 1e-7
    assert abs(res["hip"][3] - res["cpu"][3]) / abs(res["cpu"][3]) < 1e-7


def test_graph_replay_equals_plain_launches(hip_lib, monkeypatch):
    ts = np.linspace(0.0, 3e-4, 4)
    outs = []
    monkeypatch.setenv("DFX_EAGER_STEPS", "0")        # short solves would otherwise be launched eagerly in both legs
    for no_graph in ("0", "1"):
        monkeypatch.setenv("DFX_NO_GRAPH", no_graph)
        c = Case("quads", 6, True, True, seed=4, lib=None, cutoff_deg=42.0)
        y0 = c.random_state(0.05, 0.02, 5.0)
        f = c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=7)
        tree, s0 = c.solver.vjp(np.ones_like(f))
        outs.append((f, tree.geometrical_params.centroid_node_vectors, s0))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)   # same kernels, same order: bit-identical


@pytest.mark.parametrize("streams", ["1", "3"])
def test_short_solves_launched_eagerly_equal_graph_replay(hip_lib, monkeypatch, streams):
    """Solves of <= DFX_EAGER_STEPS steps skip the hipGraphs (launches of the member groups interleaved stage by stage);
    same kernels in the same per-stream order -> bit-identical fields and gradients."""
    monkeypatch.setenv("DFX_STREAMS", streams)
    ts = np.linspace(0.0, 3e-4, 4)
    outs = []
    for eager_steps in ("0", "1000"):
        monkeypatch.setenv("DFX_EAGER_STEPS", eager_steps)
        c = Case("quads", 6, True, True, seed=4, lib=None, cutoff_deg=42.0, batch=3)
        y0 = c.random_state(0.05, 0.02, 5.0)
        f = c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=7)
        trees, s0 = c.solver.vjp(np.ones_like(f))
        outs.append((f, np.stack([t.geometrical_params.centroid_node_vectors for t in trees]), s0))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_batch_members_are_independent(hip_lib):
    """3 members with different designs in one launch == 3 separate solves (bit-identical)."""
    ts = np.linspace(0.0, 3e-4, 3)
    singles, cps = [], []
    for seed in (21, 22, 23):
        c = Case("quads", 5, True, True, seed=seed, lib=None, cutoff_deg=42.0)
        f = c.solver(np.zeros((2, 25, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=8)
        tree, _ = c.solver.vjp(np.ones_like(f))
        singles.append((f, tree.geometrical_params.centroid_node_vectors))
        cps.append(c.cp)
    cb = Case("quads", 5, True, True, seed=21, lib=None, cutoff_deg=42.0, batch=3)
    fb = cb.solver(np.zeros((2, 25, 3)), ts, cps, keep_trajectory=True, steps_per_interval=8)
    trees, _ = cb.solver.vjp(np.ones_like(fb))
    for m in range(3):
        assert np.array_equal(fb[m], singles[m][0])
        assert np.array_equal(trees[m].geometrical_params.centroid_node_vectors, singles[m][1])


def test_full_size_128x128_properties(hip_lib):
    """BASELINE config C3 size (128x128 quads, contact, damping): size-independent properties.
    (a) clamped + undriven + at rest stays exactly at rest; (b) the solution is deterministic (two runs
    bit-identical); (c) the adjoint passes the dot-product test  <fields_bar, J dp> = <J^T fields_bar, dp>
    with a finite-difference directional derivative in the pulse amplitude."""
    n = 128
    c = Case("quads", n, True, True, seed=3, lib=None)
    ts = np.linspace(0.0, 2e-4, 3)
    rest = c.cp._replace(constraint_params=dict(amplitude=0.0, loading_rate=30.0, input_delay=0.0))
    f0 = c.solver(np.zeros((2, n * n, 3)), ts, rest, steps_per_interval=10)
    assert np.all(f0 == 0.0)
    fast = dict(amplitude=7.5, loading_rate=5000.0, input_delay=1e-6)
    cp = c.cp._replace(constraint_params=fast)
    f1 = c.solver(np.zeros((2, n * n, 3)), ts, cp, keep_trajectory=True, steps_per_interval=10)
    fb = np.random.default_rng(0).normal(size=f1.shape)
    tree, _ = c.solver.vjp(fb)
    f2 = c.solver(np.zeros((2, n * n, 3)), ts, cp, keep_trajectory=True, steps_per_interval=10)
    assert np.array_equal(f1, f2)
    eps = 1e-4
    fp = c.solver(np.zeros((2, n * n, 3)), ts, cp._replace(constraint_params=dict(fast, amplitude=7.5 + eps)), steps_per_interval=10)
    fm = c.solver(np.zeros((2, n * n, 3)), ts, cp._replace(constraint_params=dict(fast, amplitude=7.5 - eps)), steps_per_interval=10)
    lhs = float((fb * (fp - fm) / (2 * eps)).sum())      # all DOFs: the prescribed ones depend on the amplitude directly
    rhs = float(tree.constraint_params["amplitude"])
    assert abs(lhs - rhs) / abs(lhs) < 1e-6, (lhs, rhs)


def test_unstable_step_reports_an_error(hip_lib):
    """SURVEY 8(b) errors: NaN/Inf in the state is a non-zero status + message (the reference would return NaNs silently)."""
    c = Case("quads", 6, True, False, seed=1, lib=None)
    c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=300.0, input_delay=0.0))
    with pytest.raises(RuntimeError, match="non-finite"):
        c.solver(np.zeros((2, 36, 3)), np.linspace(0, 0.5, 3), c.cp, steps_per_interval=20)   # h = 12.5 ms >> 1/omega_max


def test_config_c2_32x32_10k_steps_forward(hip_lib, cpu_lib):
    """BASELINE config C2: 32x32 quads, nonlinear ligaments + damping, 10 000 fixed steps over 2/f, forward only.
    HIP vs the CPU port on the identical grid, plus a physical property: with the driver off again and damping on,
    the kinetic energy at the end is far below its peak."""
    ts = np.linspace(0.0, 2.0 / 30.0, 11)
    out = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 32, True, False, seed=2, lib=lib)
        out[name] = c.solver(np.zeros((2, 1024, 3)), ts, c.cp, steps_per_interval=1000)
        assert c.solver.stats["steps"] == 10000
    assert relerr(out["hip"], out["cpu"]) < 1e-8
    ke = (out["hip"][:, 1] ** 2).sum((1, 2))
    assert ke.max() > 0 and np.isfinite(out["hip"]).all()


def test_config_c4_kagome_64x64_forward_and_gradient(hip_lib, cpu_lib):
    """BASELINE config C4 lattice (64x64-cell kagome, 8 192 triangles, contact + damping, pulse): forward fields and the
    target-kinetic-energy gradient w.r.t. the three shift fields, HIP vs CPU port, on a short window."""
    from difflexmm_amd.problems import kagome_focusing_constraints, kagome_target_blocks
    res = {}
    ts = np.linspace(0.0, 3e-4, 3)
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("kagome", 64, True, True, seed=100, lib=lib, cutoff_deg=125.0)
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=5000.0, input_delay=1e-6))
        f = c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, cp, keep_trajectory=True, steps_per_interval=20)
        mid = 2 * 64 * 32
        target = np.array([mid + 2, mid + 3, mid + 4, mid + 5], dtype=np.int32)
        obj, tree, _ = c.solver.kinetic_energy_value_and_vjp(target)
        g = c.geo.vjp(c.design, tree.geometrical_params.centroid_node_vectors, tree.geometrical_params.block_centroids)
        res[name] = (f, obj, g)
    assert res["cpu"][1] > 0
    assert relerr(res["hip"][0], res["cpu"][0]) < 1e-10
    assert abs(res["hip"][1] - res["cpu"][1]) / res["cpu"][1] < 1e-10
    for a, b in zip(res["hip"][2], res["cpu"][2]):
        assert relerr(a, b) < 1e-8
    # BC index patterns of problems/kagome_focusing.py on this lattice: 2 driven blocks x 3 DOFs + 4 clamped corners
    pairs, vec, driven, clamped = kagome_focusing_constraints(c.geo, 2, 2)
    assert len(driven) == 2 and vec.sum() == 2 and len(np.unique(pairs[:, 0] * 3 + pairs[:, 1])) == len(pairs)
    assert len(kagome_target_blocks(c.geo, (2, 2), (3, 3))) == 8


def test_grid_with_its_own_step_count_per_interval(hip_lib):
    """dfx_forward_grid: a different number of steps in every output interval (power-of-two graph chunks)."""
    parity.check_trajectory_and_adjoint(None, "quads", 4, "dopri5", spi=np.array([3, 7, 1, 5]))


@pytest.mark.parametrize("lattice,n,integrator", [("quads", 4, "dopri5"), ("kagome", 3, "rk4")])
def test_grid_with_caller_chosen_step_boundaries(hip_lib, lattice, n, integrator):
    """dfx_forward_grid(step_times=...): unequal steps inside the intervals, forward and reverse."""
    parity.check_trajectory_and_adjoint(None, lattice, n, integrator, spi=np.array([3, 7, 2, 5]), own_step_times=True)


@pytest.mark.parametrize("kw", [dict(), dict(lattice="kagome", n=3, n_out=121), dict(batch=2, contact=False, nonlinear=False), dict(n=9, n_out=81)],
                         ids=["quads-contact", "kagome-contact", "linearized-batch2", "quads9-contact"])
def test_adaptive_solve_is_differentiable_as_it_stands(hip_lib, kw):
    """keep_trajectory=True without a grid -- the reference's default call under jax.grad (dynamics.py:166; problems/quads_focusing.py:565):
    ONE adaptive pass that keeps its accepted steps (dfx_forward_adaptive_keep) + the dense-output discrete adjoint, against the oracle's
    odeint restatement (forward, step boundaries) and autograd through the oracle's replay of the same steps (gradients, 1e-9)."""
    parity.check_adaptive_records_adjoint(None, **kw)


def test_adaptive_records_grow_while_the_solve_runs(hip_lib, monkeypatch):
    """The room for the accepted steps is a guess that grows (records, step boundaries, output pointers copied over): a solve started with
    room for 40 steps gives the fields and gradients of one that never had to grow."""
    res = []
    for cap in (None, "40"):
        if cap:
            monkeypatch.setenv("DFX_ADAPTIVE_CAP", cap)
        c = Case("quads", 5, True, True, seed=11, lib=None, cutoff_deg=42.0)
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
        ts = np.linspace(0, 6e-4, 31)
        s = c.solver
        s.rtol = s.atol = 1e-7
        f = s(c.random_state(0.05, 0.02, 5.0), ts, cp, keep_trajectory=True)
        assert s.stats["step_control"] == "adaptive-records" and s.stats["steps"] > 120
        tree, s0 = s.vjp(np.ones_like(f))
        res.append((f, tree.geometrical_params.centroid_node_vectors, s0, s.stats["steps"]))
    assert res[0][3] == res[1][3]
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)


def test_adaptive_grid_gradient_matches_cpu_port(hip_lib, cpu_lib, monkeypatch):
    """keep_trajectory=True without a grid: freeze the adaptive controller's accepted step boundaries, then forward +
    reverse on that grid.  The frozen solve stays within the tolerance of the adaptive one; on the SAME grid the CPU port
    gives the same fields and gradient; the CPU port's own controller picks the same grid up to the rounding
    sensitivity of the error norm (the step factor is the -1/5 power of an O(1e-8) quantity)."""
    def case(lib):
        c = Case("quads", 6, True, True, seed=9, lib=lib, cutoff_deg=42.0, batch=2)
        cps = [c.cp._replace(constraint_params=dict(amplitude=a, loading_rate=3000.0, input_delay=1e-5)) for a in (7.5, 3.0)]
        return c, cps, c.random_state(0.05, 0.02, 5.0)

    ts = np.array([0.0, 0.5e-4, 1.0e-4, 2.5e-4, 3.0e-4])
    monkeypatch.setenv("DFX_ADAPTIVE_RECORDS", "0")       # the two-pass form (what grid_refine > 1 uses)
    c, cps, y0 = case(None)
    s = c.solver
    adaptive = s(y0, ts, cps)
    frozen = s(y0, ts, cps, keep_trajectory=True)
    assert s.stats["step_control"] == "adaptive-grid"
    assert np.abs(frozen - adaptive).max() < 1e-6 * np.abs(adaptive).max()
    grid, spis = s.stats["step_times"], s.stats["steps_per_interval"]
    fb = np.random.default_rng(1).normal(size=frozen.shape)
    trees, s0 = s.vjp(fb)
    g_hip = np.stack([t.geometrical_params.centroid_node_vectors for t in trees])

    cc, cps_c, _ = case(cpu_lib)
    sc = cc.solver
    f_cpu = sc(y0, ts, cps_c, keep_trajectory=True, steps_per_interval=spis, step_times=grid)
    trees_c, s0_c = sc.vjp(fb)
    g_cpu = np.stack([t.geometrical_params.centroid_node_vectors for t in trees_c])
    assert relerr(frozen, f_cpu) < 1e-9
    assert relerr(g_hip, g_cpu) < 1e-8 and relerr(s0, s0_c) < 1e-8
    sc(y0, ts, cps_c, keep_trajectory=True)
    assert len(sc.stats["step_times"]) == len(grid) and relerr(sc.stats["step_times"], grid) < 1e-6


def test_recorded_signal_as_prescribed_displacement(hip_lib):
    """DFX_FN_TABLE: the table lives in device memory, read by the lanes that own driven DOFs."""
    parity.check_table_drive(None)


def test_several_dofs_of_one_block_share_a_time_function(hip_lib):
    """The three DOF lanes of the driven block add into the same parameter-gradient entries (atomic adds in the kernel)."""
    parity.check_several_dofs_of_one_block_share_a_time_function(None)


@pytest.mark.parametrize("lattice,n,integrator", [("quads", 5, "dopri5"), ("kagome", 3, "rk4")])
def test_reverse_sweep_reading_the_records_checkpoint(hip_lib, monkeypatch, lattice, n, integrator):
    """Third (richest) checkpoint level: the forward pass leaves EVERY stage record in the trajectory (its launches read and write
    them there instead of the ping-pong buffers), the reverse launches read the record they linearise about directly: s launches
    per step, no rebuild, no recompute.  Against the oracle, on a grid with its own step boundaries, and over graph segments."""
    monkeypatch.setenv("DFX_CHECKPOINT", "records")
    monkeypatch.setenv("DFX_EAGER_STEPS", "0")
    parity.check_trajectory_and_adjoint(None, lattice, n, integrator, spi=np.array([3, 7, 2, 5]), own_step_times=True)
    c = Case(lattice, n, True, True, seed=2, lib=None, cutoff_deg=125.0 if lattice == "kagome" else 42.0, integrator=integrator, batch=3)
    ts = np.linspace(0, 2e-4, 3)
    f = c.solver(c.random_state(0.05, 0.02, 5.0), ts, c.cp, keep_trajectory=True, steps_per_interval=300)   # several graph segments
    assert c.solver.stats["checkpoint_records"] == 1 and c.solver.stats["stage_checkpoint"] == 0
    c.solver.vjp(np.ones_like(f))
    s_launch = 6 if integrator == "dopri5" else 4
    assert abs(c.solver.adjoint_stats["launches"] / 600.0 - s_launch) < 0.2


def test_four_checkpoint_levels_give_the_same_gradient(hip_lib, monkeypatch):
    """records / stages / state / segments on one problem: same forward fields bit for bit (the same arithmetic, only what is kept
    for the reverse sweep differs), gradients equal to rounding."""
    outs = {}
    for level in ("records", "stages", "state", "segments"):
        monkeypatch.setenv("DFX_CHECKPOINT", level)
        c = Case("quads", 7, True, True, seed=12, lib=None, cutoff_deg=42.0, batch=2)
        cps = [c.cp._replace(constraint_params=dict(amplitude=a, loading_rate=3000.0, input_delay=1e-5)) for a in (7.5, -3.0)]
        ts = np.linspace(0, 3e-4, 4)
        f = c.solver(c.random_state(0.05, 0.02, 5.0), ts, cps, keep_trajectory=True, steps_per_interval=9)
        assert (c.solver.stats["checkpoint_records"], c.solver.stats["stage_checkpoint"]) == \
            {"records": (1, 0), "stages": (0, 1), "state": (0, 0), "segments": (2, 0)}[level]
        trees, s0 = c.solver.vjp(np.random.default_rng(3).normal(size=f.shape))
        outs[level] = (f, np.stack([t.geometrical_params.centroid_node_vectors for t in trees]), s0)
    for level in ("stages", "state", "segments"):
        assert np.array_equal(outs[level][0], outs["records"][0])
        assert relerr(outs[level][1], outs["records"][1]) < 1e-12 and relerr(outs[level][2], outs["records"][2]) < 1e-12


@pytest.mark.parametrize("lattice,n,integrator", [("quads", 5, "dopri5"), ("kagome", 3, "rk4")])
def test_reverse_sweep_recomputing_one_output_interval_at_a_time(hip_lib, cpu_lib, monkeypatch, lattice, n, integrator):
    """Fourth level ("segments"): the forward pass keeps only its outputs; the reverse sweep re-runs the forward pass of one output
    interval at a time (from the resident output row, with the records checkpoint for that interval) and reverses it.  Memory is
    independent of the horizon.  Against the oracle on a grid with its own step boundaries, and against the CPU port over intervals
    that span several launch segments, 3 members in 2 groups."""
    monkeypatch.setenv("DFX_CHECKPOINT", "segments")
    monkeypatch.setenv("DFX_STREAMS", "2")
    parity.check_trajectory_and_adjoint(None, lattice, n, integrator, spi=np.array([3, 7, 2, 5]), own_step_times=True)
    res = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case(lattice, n, True, True, seed=2, lib=lib, cutoff_deg=125.0 if lattice == "kagome" else 42.0, integrator=integrator, batch=3)
        ts = np.linspace(0, 3e-4, 4)
        y0 = c.random_state(0.05, 0.02, 5.0)
        f = c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=np.array([300, 40, 270]))
        if lib is None:
            assert c.solver.stats["checkpoint_records"] == 2
        fb = np.random.default_rng(4).normal(size=f.shape)
        trees, s0 = c.solver.vjp(fb)
        res[name] = (f, np.stack([t.geometrical_params.centroid_node_vectors for t in trees]), s0)
    assert relerr(res["hip"][0], res["cpu"][0]) < 1e-10
    assert relerr(res["hip"][1], res["cpu"][1]) < 1e-8 and relerr(res["hip"][2], res["cpu"][2]) < 1e-8


@pytest.mark.parametrize("stage_checkpoint", ["0", "1"])
@pytest.mark.parametrize("lattice,n,integrator", [("quads", 5, "dopri5"), ("kagome", 3, "rk4")])
def test_reverse_sweep_with_and_without_stage_checkpoint(hip_lib, monkeypatch, stage_checkpoint, lattice, n, integrator):
    """Two ways through the reverse sweep: stage accelerations of every step kept by the forward pass (records rebuilt
    elementwise, s launches per step) or only the step states (records recomputed, 2s - 1 launches per step).  The engine
    picks the first whenever it fits in HBM; both must agree with the oracle."""
    monkeypatch.setenv("DFX_STAGE_CHECKPOINT", stage_checkpoint)
    parity.check_trajectory_and_adjoint(None, lattice, n, integrator, spi=np.array([3, 7, 2, 5]), own_step_times=True)
    c = Case(lattice, n, True, True, seed=2, lib=None, cutoff_deg=125.0 if lattice == "kagome" else 42.0, integrator=integrator, batch=3)
    ts = np.linspace(0, 2e-4, 3)
    f = c.solver(c.random_state(0.05, 0.02, 5.0), ts, c.cp, keep_trajectory=True, steps_per_interval=300)   # several graph segments
    assert c.solver.stats["stage_checkpoint"] == int(stage_checkpoint)
    c.solver.vjp(np.ones_like(f))
    s_launch = 6 if integrator == "dopri5" else 4
    per_step = c.solver.adjoint_stats["launches"] / 600.0
    assert abs(per_step - (s_launch if stage_checkpoint == "1" else 2 * s_launch - 1)) < 0.2


@pytest.mark.parametrize("streams,stage_checkpoint", [("3", "1"), ("3", "0"), ("2", "1")])
def test_member_groups_on_concurrent_streams_match_cpu_port(hip_lib, cpu_lib, monkeypatch, streams, stage_checkpoint):
    """5 members split into groups that advance on their own HIP streams with their own graphs (what bench.py runs with 16
    members in 2 groups): fields and gradients of every member equal the CPU port's, in both reverse-sweep modes."""
    monkeypatch.setenv("DFX_STREAMS", streams)
    monkeypatch.setenv("DFX_STAGE_CHECKPOINT", stage_checkpoint)
    res = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 8, True, True, seed=21, lib=lib, cutoff_deg=42.0, batch=5)
        cps = [c.cp._replace(constraint_params=dict(amplitude=a, loading_rate=3000.0, input_delay=1e-5)) for a in (7.5, 3.0, -4.0, 6.0, 1.0)]
        ts = np.linspace(0, 3e-4, 4)
        y0 = c.random_state(0.05, 0.02, 5.0)
        f = c.solver(y0, ts, cps, keep_trajectory=True, steps_per_interval=270)     # > 256 steps: two graph segments per interval
        if name == "hip":
            assert c.solver.stats["streams"] == int(streams) and c.solver.stats["stage_checkpoint"] == int(stage_checkpoint)
        fb = np.random.default_rng(5).normal(size=f.shape)
        trees, s0 = c.solver.vjp(fb)
        res[name] = (f, np.stack([t.geometrical_params.centroid_node_vectors for t in trees]),
                     np.array([t.constraint_params["amplitude"] for t in trees]), s0)
    assert relerr(res["hip"][0], res["cpu"][0]) < 1e-10
    assert relerr(res["hip"][1], res["cpu"][1]) < 1e-8
    assert relerr(res["hip"][2], res["cpu"][2]) < 1e-8
    assert relerr(res["hip"][3], res["cpu"][3]) < 1e-8


@pytest.mark.parametrize("dual_chain", ["0", "1"])
def test_single_system_reverse_sweep_on_two_chains(hip_lib, cpu_lib, monkeypatch, dual_chain):
    """One system, state checkpoint only: the recompute launches of step n-1 overlap the reverse launches of step n on a
    second stream (double-buffered stage records) -- same gradient as the single chain and as the CPU port."""
    monkeypatch.setenv("DFX_DUAL_CHAIN", dual_chain)
    monkeypatch.setenv("DFX_STAGE_CHECKPOINT", "0")
    res = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 8, True, True, seed=22, lib=lib, cutoff_deg=42.0)
        cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
        ts = np.linspace(0, 3e-4, 4)
        f = c.solver(c.random_state(0.05, 0.02, 5.0), ts, cp, keep_trajectory=True, steps_per_interval=270)
        fb = np.random.default_rng(6).normal(size=f.shape)
        tree, s0 = c.solver.vjp(fb)
        res[name] = (f, tree.geometrical_params.centroid_node_vectors, s0)
    for a, b, tol in zip(res["hip"], res["cpu"], (1e-10, 1e-8, 1e-8)):
        assert relerr(a, b) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["records", "state", "segments"])
def test_design_subset_of_the_gradient_equals_the_full_set(hip_lib, monkeypatch, level):
    """Asking only for what a design reaches (node vectors, void angles, inertia -- plus state0 here) runs the reverse stage in its
    build without per-ligament gradients, which keeps lambda / Ybar as (q, v) pairs (DevCtx::lam_pairs); asking for every leaf runs
    the other build with the scalar layout.  Same objective, same numbers for the common leaves, state0 (read back from lambda) included."""
    monkeypatch.setenv("DFX_CHECKPOINT", level)
    c = Case("quads", 7, True, True, seed=5, lib=None, cutoff_deg=42.0, batch=2)
    cps = [c.cp._replace(constraint_params=dict(amplitude=a, loading_rate=3000.0, input_delay=1e-5)) for a in (7.5, -3.0)]
    ts = np.linspace(0, 3e-4, 4)
    target = np.array([16, 17, 23, 24], dtype=np.int32)
    y0 = c.random_state(0.05, 0.02, 5.0)
    sub = ("centroid_node_vectors", "void_angle0", "inertia", "state0")
    c.solver(y0, ts, cps, keep_trajectory=True, steps_per_interval=9, want_fields=False)
    o1, g1, _ = c.solver.engine.kinetic_value_and_grad(target, which=sub)
    g1 = {k: np.array(v) for k, v in g1.items()}
    c.solver(y0, ts, cps, keep_trajectory=True, steps_per_interval=9, want_fields=False)
    o2, g2, _ = c.solver.engine.kinetic_value_and_grad(target)
    assert np.array_equal(o1, o2) and np.all(o1 > 0)
    for k in sub:
        assert np.abs(g1[k]).max() > 0 and relerr(g1[k], np.array(g2[k])) < 1e-12, k


@pytest.mark.parametrize("lattice,n", [("quads", 9), ("kagome", 5)])
def test_rhs_and_vjp_with_rotations_in_every_quadrant(hip_lib, lattice, n):
    """The stage records keep sin(theta/2) only; cos(theta/2) = +-sqrt(1 - sin^2) with the sign from the quadrant of theta/2
    (dfx_physics.h: half_cos).  Block rotations drawn with a standard deviation of 3 rad put theta/2 in all four quadrants and beyond
    one turn: one RHS and every VJP against autograd through the oracle (digits are lost only within a fraction of a degree of
    |theta| = pi, hence 1e-10 instead of 1e-12)."""
    parity.check_rhs_and_vjp(None, lattice, n, True, False, seed=11, scale_th=3.0, rtol=1e-10)


def test_contact_culling_bound_with_some_ligaments_beyond_it(hip_lib):
    """Angle contact with the cutoff a little below the smallest undeformed void angle: the per-member culling bound
    (pack_params: kappa_safe = min(phi_lo - cutoff, pi - phi_hi)) is then a few degrees, the random rotations put some ligaments
    beyond it (they load their void angles and may touch) and leave the others culled (exact zeros without loading them).
    RHS, y_bar and every parameter gradient, the contact constants included, against autograd through the oracle."""
    from difflexmm_amd import geometry as geo
    from .common import Case
    c = Case("quads", 9, True, True, seed=3, lib=None, cutoff_deg=42.0)
    phi = geo.void_angles0(c.cnv, c.bonds)
    lo = float(np.min(phi)) * 180 / np.pi
    errs = parity.check_rhs_and_vjp(None, "quads", 9, True, True, seed=3, scale_th=0.06, cutoff_deg=lo - 2.0)
    assert errs["contact"] < 1e-12
