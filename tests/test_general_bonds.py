"""Nodes that carry more than one ligament (jax_md.smap.bond takes any bond list, difflexmm/energy.py:179-197; the lattice
generators never produce one, rounds 1-2 refused them): the first ligament of a node is a lane's regular one, the others are walked
in a per-lane loop (Plan::ovf_*).  Quads with three kinds of extra ligaments -- a node that gets a second and a third one, a node
whose FIRST ligament is an extra one of its partner, and an extra ligament between two nodes that both already carry one -- against
the torch oracle: one RHS + every VJP, then a short trajectory + discrete adjoint.  CPU port here, HIP engine in test_gpu_general_bonds."""
import numpy as np
import pytest

from . import parity

# quads n = 4: block b has nodes 4b .. 4b+3 (0:+x, 1:+y, 2:-x, 3:-y); interior nodes carry one lattice ligament, rim nodes none
EXTRA = np.array([[4 * 5 + 0, 4 * 10 + 3],      # node (5, +x) (bonded to block 6) gets a second ligament, to node (10, -y) (bonded to block 6)
                  [4 * 5 + 0, 4 * 15 + 0],      # ... and a third one, to the rim node (15, +x): that node's FIRST ligament
                  [4 * 0 + 2, 4 * 9 + 1],       # rim node (0, -x): first ligament; node (9, +y) (bonded to block 13): second
                  [4 * 12 + 1, 4 * 3 + 0]])     # two rim nodes: a plain additional bond (both first ligaments)


def check_rhs(lib, contact):
    errs = parity.check_rhs_and_vjp(lib, "quads", 4, True, contact, seed=4, extra_bonds=EXTRA, cutoff_deg=80.0, rtol=1e-11)
    assert errs["rhs"] < 1e-12 and errs["k"] < 1e-11 and errs["refv"] < 1e-11


def check_trajectory(lib):
    parity.check_trajectory_and_adjoint(lib, "quads", 4, "dopri5", contact=True, extra_bonds=EXTRA, spi=6, n_out=4)


SCRAMBLED = [("quads", 5, 80.0, 1), ("quads", 5, 80.0, 2), ("kagome", 3, 150.0, 3)]


def check_scrambled(lib, lattice, n, cut, sc):
    """An arbitrary bond list made from the lattice's: a quarter of the ligaments removed, random order, half of them with swapped ends
    (reference vector negated), three ligaments between random nodes (tests/common.py Case(scramble=...)); contact engaged."""
    for contact in (False, True):
        parity.check_rhs_and_vjp(lib, lattice, n, True, contact, seed=10 + sc, scramble=sc, cutoff_deg=cut, rtol=1e-11)
    parity.check_trajectory_and_adjoint(lib, lattice, n, "dopri5", contact=True, scramble=sc, spi=6, n_out=4)


@pytest.mark.parametrize("lattice,n,cut,sc", SCRAMBLED)
def test_scrambled_bond_lists_cpu_port(cpu_lib, lattice, n, cut, sc):
    check_scrambled(cpu_lib, lattice, n, cut, sc)


def test_extra_ligaments_rhs_and_vjp_cpu_port(cpu_lib):
    check_rhs(cpu_lib, False)
    check_rhs(cpu_lib, True)


def test_extra_ligaments_trajectory_and_adjoint_cpu_port(cpu_lib):
    check_trajectory(cpu_lib)


def test_response_data_counts_every_ligament_once(cpu_lib):
    from .common import Case
    c = Case("quads", 4, True, False, seed=2, lib=cpu_lib, extra_bonds=EXTRA)
    c.cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    ts = np.linspace(0, 2e-4, 3)
    c.solver(np.zeros((2, 16, 3)), ts, c.cp, steps_per_interval=8)
    out = c.solver.engine.response_data()
    e = out["strain_energy_stretch"][0] + out["strain_energy_shear"][0] + out["strain_energy_bending"][0]     # (T, n_bonds)
    assert e.shape == (3, len(c.bonds)) and np.all(e[0] < 1e-25) and np.all(e[-1][-len(EXTRA):] > 1e-12)
    # the potential energy of the final configuration = the sum over ALL ligaments, extra ones included
    flat = c.solver._flatten(c.cp)
    c.solver.engine.set_params(**{k: v[None] for k, v in flat.items()})
    fields = c.solver(np.zeros((2, 16, 3)), ts, c.cp, steps_per_interval=8)
    total = c.solver.engine.energy(fields[-1, 0][None])[0]
    assert abs(e[-1].sum() - total) < 1e-12 * total
