"""The reference's `scripts/pulse_RS.py` on the engine: a rotated-squares domain (20 x 10 cells), no constraints, the second column of
blocks loaded with the sech^2 tanh force pulse, the default adaptive Dormand-Prince call, timed twice like the script (the second call
re-uses the engine handle: the counterpart of calling the jitted solver again).

    python examples/pulse_rotated_squares.py [--n1 20] [--n2 10] [--out solution.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import difflexmm_amd as dm  # noqa: E402
from difflexmm_amd import energy, geometry, loading  # noqa: E402
from difflexmm_amd.dynamics import setup_dynamic_solver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=20)
    ap.add_argument("--n2", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    squares = geometry.RotatedSquareGeometry(n1_cells=a.n1, n2_cells=a.n2, bond_length=0.1)
    initial_angle = 0.35
    k_stretch, k_shear, k_rot, density = 1.0, 0.33, 0.0075, 1.0
    cnv = squares.centroid_node_vectors(initial_angle)
    inertia = geometry.compute_inertia(cnv, density)
    potential_energy = energy.build_strain_energy(squares.bond_connectivity(), energy.ligament_energy)
    loaded = np.array([[squares.n1_blocks * i + 1, 0] for i in range(squares.n2_blocks)])
    amplitude, sharpness = 0.3, 4.0       # loading(t) = 2 A / s^2 sech^2(t / s - 3) tanh(3 - t / s)
    solve_dynamics = setup_dynamic_solver(squares, potential_energy, loaded_block_DOF_pairs=loaded,
                                          loading_fn=loading.Sech2Tanh(amplitude=amplitude, width=sharpness),
                                          constrained_block_DOF_pairs=np.zeros((0, 2), dtype=np.int64))
    control_params = dm.ControlParams(
        dm.GeometricalParams(squares.block_centroids(initial_angle), cnv),
        dm.MechanicalParams(dm.LigamentParams(k_stretch, k_shear, k_rot, squares.reference_bond_vectors()), density, inertia))
    timepoints = np.linspace(0.0, float(squares.n1_blocks), 100)
    state0 = np.zeros((2, squares.n_blocks, 3))
    for label in ("first call", "second call"):
        t0 = time.perf_counter()
        solution = solve_dynamics(state0, timepoints, control_params)
        print(f"Solution time ({label}): {time.perf_counter() - t0:.3f} s   "
              f"[{solve_dynamics.stats['steps']} accepted steps, max |u| = {np.abs(solution[:, 0]).max():.4f}]")
    if a.out:
        np.savez_compressed(a.out, block_centroids=squares.block_centroids(initial_angle), centroid_node_vectors=cnv,
                            bond_connectivity=squares.bond_connectivity(), timepoints=timepoints, fields=solution)


if __name__ == "__main__":
    main()
