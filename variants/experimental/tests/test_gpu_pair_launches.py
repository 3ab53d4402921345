"""-m gpu: the pair launches (two RK stages per launch on lattice windows, dfx_pair.h) forced on with DFX_PAIR=1, against the
one-stage launches and against the torch oracle.  Lattices larger than one window in both directions (windows with rings on every
side, lattice edges, clipped last windows), both window heights, quads and kagome, every checkpoint level the forward pair serves."""
import os

import numpy as np
import pytest

from tests.common import Case, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_launch_per_stage(monkeypatch):
    """These are A/B tests between builds of the STAGE launches: the persistent stage loop (dfx_persist.h), which would serve lattices of
    this size by default, stays out of both arms (tests/test_gpu_persistent.py is its own A/B)."""
    monkeypatch.setenv("DFX_PERSIST", "0")


def _solve(c, ts, spi, target, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        fields = c.solver(np.zeros((2, c.geo.n_blocks, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=spi)
        obj, raw = c.solver.kinetic_energy_value_and_raw(target)
        return (fields.copy(), float(np.atleast_1d(obj)[0]), {k: np.array(v) for k, v in raw.items()}, dict(c.solver.adjoint_stats),
                c.solver.stats["launches"])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


FAST = dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5)


@pytest.mark.parametrize("lattice,n,rows", [("quads", 37, "16"), ("quads", 37, "8"), ("kagome", 21, "16"), ("quads", 12, "8")])
def test_pair_launches_equal_stage_launches(experimental_lib, lattice, n, rows):
    c = Case(lattice, n, True, True, seed=21, cutoff_deg=42.0 if lattice == "quads" else 125.0)
    c.cp = c.cp._replace(constraint_params=FAST)
    ts = np.linspace(0.0, 3e-4, 4)
    mid = c.geo.n_blocks // 2
    target = np.array([mid + 1, mid + 2], dtype=np.int32)
    ref = _solve(c, ts, 7, target, {"DFX_PAIR": "0"})
    for level in ("records", "segments"):
        out = _solve(c, ts, 7, target, {"DFX_PAIR": "1", "DFX_PAIR_ROWS": rows, "DFX_CHECKPOINT": level})
        assert c.solver.stats["launches"] < 0.6 * ref[4]                    # the pair launches really ran: half the forward launches
        if level == "records":
            assert out[3]["launches"] < 0.6 * ref[3]["launches"]            # ... and half the reverse launches
        assert relerr(out[0], ref[0]) < 1e-12 and abs(out[1] - ref[1]) < 1e-12 * abs(ref[1])
        for k in ref[2]:
            assert relerr(out[2][k], ref[2][k]) < 1e-10, (level, k)
    # forward pairs at the levels whose reverse sweep keeps the stage launches
    for level in ("stages", "state"):
        out = _solve(c, ts, 7, target, {"DFX_PAIR": "f", "DFX_PAIR_ROWS": rows, "DFX_CHECKPOINT": level})
        assert relerr(out[0], ref[0]) < 1e-12
        for k in ref[2]:
            assert relerr(out[2][k], ref[2][k]) < 1e-10, (level, k)


def test_pair_launches_match_the_oracle(experimental_lib):
    """20 x 20 quads (2 x 2 windows), contact engaged: fields of a 24-step solve against the oracle's fixed-grid solver."""
    c = Case("quads", 20, True, True, seed=5, cutoff_deg=42.0)
    c.cp = c.cp._replace(constraint_params=FAST)
    ts = np.linspace(0.0, 2.4e-4, 3)
    os.environ["DFX_PAIR"] = "1"
    try:
        fields = c.solver(np.zeros((2, 400, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=12)
    finally:
        os.environ.pop("DFX_PAIR")
    import torch
    lv = dict(loading_rate=torch.tensor(3000.0, dtype=torch.float64), input_delay=torch.tensor(1e-5, dtype=torch.float64))
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=12)
    assert relerr(fields, osol(np.zeros((2, 400, 3)), ts, c.oracle_cp(lv)).numpy()) < 1e-10


@pytest.mark.parametrize("n", [7, 21])
def test_packed_triangles_equal_the_quad_mapping(hip_lib, n):
    """3-node blocks: five triangles per 16 lanes (lane_pos<3>, DPP row moves) against one triangle per quad (the mapping of rounds
    1-2, DFX_PACK3=0): fields, objective, every gradient the records build accumulates.  n = 7: 98 blocks = two and a half
    workgroups with a ragged last row; n = 21: 882 blocks."""
    c = Case("kagome", n, True, True, seed=8, cutoff_deg=125.0)
    c.cp = c.cp._replace(constraint_params=FAST)
    ts = np.linspace(0.0, 3e-4, 4)
    mid = c.geo.n_blocks // 2
    target = np.array([mid + 1, mid + 2], dtype=np.int32)
    ref = _solve(c, ts, 7, target, {"DFX_PACK3": "0", "DFX_CHECKPOINT": "records"})
    for level in ("records", "state"):
        out = _solve(c, ts, 7, target, {"DFX_PACK3": "1", "DFX_CHECKPOINT": level})
        assert relerr(out[0], ref[0]) < 1e-12 and abs(out[1] - ref[1]) < 1e-12 * abs(ref[1])
        for k in ref[2]:
            assert relerr(out[2][k], ref[2][k]) < 1e-10, (level, k)
    assert np.abs(ref[2]["centroid_node_vectors"]).max() > 0 and ref[1] > 0
