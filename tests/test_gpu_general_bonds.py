"""-m gpu: nodes with more than one ligament on the HIP engine (tests/test_general_bonds.py has the cases and the CPU-port run)."""
import numpy as np
import pytest

from . import test_general_bonds as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lattice,n,cut,sc", G.SCRAMBLED)
def test_scrambled_bond_lists_hip(hip_lib, lattice, n, cut, sc):
    G.check_scrambled(None, lattice, n, cut, sc)


def test_extra_ligaments_rhs_and_vjp_hip(hip_lib):
    G.check_rhs(None, False)
    G.check_rhs(None, True)


def test_extra_ligaments_trajectory_and_adjoint_hip(hip_lib):
    G.check_trajectory(None)


def test_response_data_and_every_checkpoint_level_hip(hip_lib, cpu_lib, monkeypatch):
    """Per-ligament energies incl. the extra ones, and the design-subset gradient at every checkpoint level against the CPU port."""
    from .common import Case, relerr
    out = {}
    for name, lib in (("hip", None), ("cpu", cpu_lib)):
        c = Case("quads", 4, True, True, seed=2, lib=lib, cutoff_deg=80.0, extra_bonds=G.EXTRA)
        c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
        ts = np.linspace(0, 2e-4, 3)
        levels = ("records", "stages", "state", "segments") if lib is None else ("records",)
        for level in levels:
            monkeypatch.setenv("DFX_CHECKPOINT", level)
            c.solver(np.zeros((2, 16, 3)), ts, c.cp, keep_trajectory=True, steps_per_interval=8)
            obj, raw = c.solver.kinetic_energy_value_and_raw(np.array([5, 6], dtype=np.int32))
            out[(name, level)] = (float(np.atleast_1d(obj)[0]), {k: np.array(v) for k, v in raw.items()})
        out[(name, "resp")] = c.solver.engine.response_data()
    ref = out[("cpu", "records")]
    for level in ("records", "stages", "state", "segments"):
        got = out[("hip", level)]
        assert abs(got[0] - ref[0]) < 1e-11 * abs(ref[0])
        for k in ref[1]:
            assert relerr(got[1][k], ref[1][k]) < 1e-9, (level, k)
    for k in ("strain_energy_stretch", "strain_energy_shear", "strain_energy_bending"):
        assert relerr(out[("hip", "resp")][k], out[("cpu", "resp")][k]) < 1e-11
