"""How much of a launch-bound stage launch is the tail of the few lanes that evaluate time functions (driven / clamped blocks)?
One 128x128 system, forward only, 500 steps: the C3 boundary conditions against the same lattice with no constrained DOF at all."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import difflexmm_amd as dm
from difflexmm_amd import energy as E, geometry as G, loading as L
from difflexmm_amd.dynamics import setup_dynamic_solver
from difflexmm_amd.problems import quads_focusing_constraints
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = G.QuadGeometry(n, n, 15.0, 2.25)
bonds = g.bond_connectivity()
en = E.combine_block_energies(E.build_strain_energy(bonds, E.ligament_energy), E.build_contact_energy(bonds))
design = g.get_design_from_rotated_square(25 * math.pi / 180)
cen, cnv = g.geometry_from_design(*design)
rho = 6.18e-9
damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * 225 * 1.19)] * 2 + [2 * math.sqrt(0.02175026 * rho * 15.0 ** 4 * 1.5)]) * np.ones((n * n, 1))
cp = dm.ControlParams(dm.GeometricalParams(cen, cnv), dm.MechanicalParams(dm.LigamentParams(120.0, 1.19, 1.5, g.reference_bond_vectors()), rho, None, damping,
                      dm.ContactParams(-15 * math.pi / 180, -10 * math.pi / 180, 1.5)), constraint_params=dict(amplitude=7.5, loading_rate=30.0, input_delay=0.0))
pairs, vec, _, _ = quads_focusing_constraints(g, 2, "left", 0, 2)
ts = np.arange(3) * 250 * (2.0 / 30 / 50000)
for label, kw in (("C3 boundary conditions", dict(constrained_block_DOF_pairs=pairs, constrained_DOFs_fn=L.Pulse(vec))), ("no constrained DOF", {})):
    s = setup_dynamic_solver(g, en, damped_blocks=np.arange(n * n), batch=B, **kw)
    for rep in range(3):
        s(np.zeros((2, n * n, 3)) + (1e-3 if not kw else 0.0), ts, [cp] * B if B > 1 else cp, steps_per_interval=250, want_fields=False)
    st = s.stats
    print(f"{label}: {B} x {n}x{n}, 500 steps forward: {st['kernel_ms']:.2f} ms, {1e3 * st['kernel_ms'] / st['launches']:.2f} us per launch ({st['launches']} launches)")
    s.engine.close()
