"""scripts/pulse_RS.py of the reference as written -- 40 x 20 rotated squares, no constraints, no damping, sech^2 tanh force pulse on the
second column, default adaptive odeint -- on the engine API, against the golden of the oracle (tests/golden/pulse_rs.npz)."""
import os

import numpy as np

import difflexmm_amd as dm
from difflexmm_amd import energy as en_mod
from difflexmm_amd import geometry as geo_mod
from difflexmm_amd import loading as ld
from difflexmm_amd.dynamics import setup_dynamic_solver

from .common import relerr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check(lib):
    gold = np.load(os.path.join(GOLD, "pulse_rs.npz"))
    g = geo_mod.RotatedSquareGeometry(n1_cells=int(gold["n1_cells"]), n2_cells=int(gold["n2_cells"]), bond_length=0.1)
    ang = 0.35
    cnv = g.centroid_node_vectors(ang)
    inertia = geo_mod.compute_inertia(cnv, 1.0)
    assert relerr(inertia, gold["inertia"]) < 1e-14
    energy = en_mod.build_strain_energy(g.bond_connectivity(), en_mod.ligament_energy)
    loaded = np.array([[g.n1_blocks * i + 1, 0] for i in range(g.n2_blocks)])
    # the script's loading(state, t) = 2 A / s^2 sech^2(t / s - 3) tanh(3 - t / s), A = 0.3, s = 4
    solver = setup_dynamic_solver(g, energy, loaded_block_DOF_pairs=loaded, loading_fn=ld.Sech2Tanh(amplitude=0.3, width=4.0),
                                  constrained_block_DOF_pairs=np.zeros((0, 2), dtype=np.int64), _lib=lib)
    cp = dm.ControlParams(dm.GeometricalParams(g.block_centroids(ang), cnv),
                          dm.MechanicalParams(dm.LigamentParams(1.0, 0.33, 0.0075, g.reference_bond_vectors()), 1.0, inertia))
    f = solver(np.zeros((2, g.n_blocks, 3)), gold["timepoints"], cp)           # default call: adaptive, rtol = atol = 1e-8
    assert solver.stats["step_control"] == "adaptive"
    assert abs(solver.stats["steps"] - int(gold["accepted"])) <= 1
    # same controller and tableau -> same step sequence; what is left is rounding (the pulse travels 40 blocks, |q| ~ 0.2)
    assert relerr(f[:, 0], gold["fields"][:, 0]) < 1e-9 and relerr(f[:, 1], gold["fields"][:, 1]) < 1e-9
    return f
