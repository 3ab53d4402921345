"""-m gpu: the tile kernels (every ligament evaluated once, three lanes per block on wave-private tiles of 7 x 3 blocks, dfx_tile.h)
forced on with DFX_TILE=1, against the slot kernels and against the torch oracle.  Lattices larger than one tile in both directions
with clipped last tiles (37 = 5 x 7 + 2 columns, 12 x 3 + 1 rows; an odd number of tiles: the second wave of the last workgroup has
none), quads (k = 1 partner straight above) and kagome (diagonal: dc1 = -1), the checkpoint levels whose reverse sweep the tile build
serves."""
import os

import numpy as np
import pytest

from tests.common import Case, relerr
from test_gpu_pair_launches import FAST, _solve

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_launch_per_stage(monkeypatch):
    """These are A/B tests between builds of the STAGE launches: the persistent stage loop (dfx_persist.h), which would serve lattices of
    this size by default, stays out of both arms (tests/test_gpu_persistent.py is its own A/B)."""
    monkeypatch.setenv("DFX_PERSIST", "0")


def _case(lattice, n, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)            # the lane tables are built when the handle is created
    try:
        c = Case(lattice, n, True, True, seed=21, cutoff_deg=42.0 if lattice == "quads" else 125.0)
        c.solver                      # noqa: B018  (creates the engine)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.cp = c.cp._replace(constraint_params=FAST)
    return c


@pytest.mark.parametrize("lattice,n", [("quads", 37), ("kagome", 21), ("quads", 12), ("kagome", 7), ("quads", 15)])
def test_tile_kernels_equal_slot_kernels(experimental_lib, lattice, n):
    ts = np.linspace(0.0, 3e-4, 4)
    ref_c = _case(lattice, n, {"DFX_TILE": "0"})
    mid = ref_c.geo.n_blocks // 2
    target = np.array([mid + 1, mid + 2], dtype=np.int32)
    ref = _solve(ref_c, ts, 7, target, {"DFX_CHECKPOINT": "records"})
    assert ref_c.solver.stats["tile_kernels"] == 0 and ref[3]["tile_kernels"] == 0
    c = _case(lattice, n, {"DFX_TILE": "1"})
    for level in ("records", "segments", "state"):
        out = _solve(c, ts, 7, target, {"DFX_CHECKPOINT": level})
        assert c.solver.stats["tile_kernels"] == 1 and out[3]["tile_kernels"] == 1       # the tile kernels really ran, both directions
        assert relerr(out[0], ref[0]) < 1e-12 and abs(out[1] - ref[1]) < 1e-12 * abs(ref[1]), level
        for k in ref[2]:
            assert relerr(out[2][k], ref[2][k]) < 1e-10, (level, k)
    assert np.abs(ref[2]["centroid_node_vectors"]).max() > 0 and np.abs(ref[2]["void_angle0"]).max() > 0 and ref[1] > 0


def test_tile_kernels_match_the_oracle(experimental_lib):
    """20 x 20 quads (2 x 3 tiles), contact engaged: fields of a 24-step solve against the oracle's fixed-grid solver."""
    c = _case("quads", 20, {"DFX_TILE": "1"})
    ts = np.linspace(0.0, 2.4e-4, 3)
    fields = c.solver(np.zeros((2, 400, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=12)
    import torch
    lv = dict(loading_rate=torch.tensor(3000.0, dtype=torch.float64), input_delay=torch.tensor(1e-5, dtype=torch.float64))
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=12)
    assert relerr(fields, osol(np.zeros((2, 400, 3)), ts, c.oracle_cp(lv)).numpy()) < 1e-10


def test_per_stage_builds_equal_generic_builds(hip_lib):
    """The per-stage / common-shape builds of the slot kernels (what an ensemble that fills the chip launches: DESIGN.md section 4) against
    their generic builds on the same problem: 8 designs of 64 x 64 quads (2 048 waves per launch), records and segments levels --
    fields 1e-13, every gradient the records build accumulates 1e-11 -- and dfx_stats says which build ran."""
    ts = np.linspace(0.0, 3e-4, 4)
    out = {}
    for name, env in (("generic", {"DFX_STAGE_BUILDS": "0"}), ("plain", {"DFX_WT": "0"}), ("hot", {})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            c = Case("quads", 64, True, True, seed=21, cutoff_deg=42.0, batch=8, per_bond_k=False)
            c.solver
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        c.cp = [c.cp._replace(constraint_params=dict(FAST, amplitude=7.5 * (1 + 0.02 * m))) for m in range(8)]
        mid = c.geo.n_blocks // 2
        target = np.array([mid + 1, mid + 2], dtype=np.int32)
        res = []
        for level in ("records", "segments"):
            r = _solve(c, ts, 7, target, {"DFX_CHECKPOINT": level})
            res.append(r)
            want = 2 if name == "hot" else 0
            assert c.solver.stats["tile_kernels"] == want and r[3]["tile_kernels"] == want, (name, level, c.solver.stats["tile_kernels"])
        out[name] = res
    for name in ("plain", "hot"):
        for a, b in zip(out[name], out["generic"]):
            assert relerr(a[0], b[0]) < 1e-13 and abs(a[1] - b[1]) < 1e-12 * abs(b[1]), name
            for k in b[2]:
                assert relerr(a[2][k], b[2][k]) < 1e-11, (name, k)
    assert np.abs(out["hot"][0][2]["centroid_node_vectors"]).max() > 0
