"""Boundary-condition patterns, target-block lists and objectives of the reference's focusing problems
(the callers of the hot path), restated in NumPy:

* quads:  ``problems/quads_focusing.py:104-209`` (driven edge blocks + clamped corners), ``:447-451`` (target)
* kagome: ``problems/kagome_focusing.py:96-172``, ``:403-407``
* objective ``target_kinetic_energy``: ``problems/quads_focusing.py:453-467`` with ``energy.py:494-499``
"""
from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np

from . import energy as E
from . import loading as L
from .dynamics import setup_dynamic_solver
from .geometry import KagomeGeometry, QuadGeometry, compute_inertia
from .utils import (ContactParams, ControlParams, GeometricalParams, LigamentParams, MechanicalParams, SolutionData)


def _tile3(blocks):
    blocks = np.asarray(blocks, dtype=np.int64)
    n = len(blocks)
    return np.stack([np.tile(blocks, 3), np.repeat(np.arange(3), n)], 1)


def quads_focusing_constraints(geometry: QuadGeometry, n_excited_blocks: int, loaded_side: str = "left",
                               input_shift: int = 0, n_blocks_clamped_corner: int = 2):
    """(constrained_block_DOF_pairs, loading_vector, driven_blocks, clamped_blocks) exactly as
    problems/quads_focusing.py:104-209 builds them (same ordering of the pairs)."""
    n1, n2, nb = geometry.n1_blocks, geometry.n2_blocks, geometry.n_blocks
    ne, sh, nc = n_excited_blocks, input_shift, n_blocks_clamped_corner
    if loaded_side == "left":
        blocks = np.arange((n2 - ne) // 2 + sh, (n2 + ne) // 2 + sh) * n1
        dofs = [0] * ne + [1] * ne + [2] * ne
    elif loaded_side == "right":
        blocks = np.arange((n2 - ne) // 2 + sh, (n2 + ne) // 2 + sh) * n1 + (n1 - 1)
        dofs = [0] * ne + [1] * ne + [2] * ne
    elif loaded_side == "bottom":
        blocks = np.arange((n1 - ne) // 2 + sh, (n1 + ne) // 2 + sh)
        dofs = [1] * ne + [0] * ne + [2] * ne
    elif loaded_side == "top":
        blocks = np.arange((n1 - ne) // 2 + sh, (n1 + ne) // 2 + sh) + n1 * (n2 - 1)
        dofs = [1] * ne + [0] * ne + [2] * ne
    else:
        raise ValueError(f"Unknown loaded_side: {loaded_side}. Should be either 'left', 'right', 'bottom' or 'top'.")
    driven = np.stack([np.tile(blocks, 3), np.array(dofs)], 1)
    bl = np.concatenate([np.arange(0, nc), [i * n1 for i in range(1, nc)]])
    br = np.concatenate([np.arange(n1 - nc, n1), [(i + 1) * n1 - 1 for i in range(1, nc)]])
    tr = np.concatenate([np.arange(nb - nc, nb), [nb - i * n1 - 1 for i in range(1, nc)]])
    tl = np.concatenate([np.arange(nb - n1, nb - n1 + nc), [nb - n1 - i * n1 for i in range(1, nc)]])
    clamped = [_tile3(c) for c in (bl, br, tr, tl)]
    pairs = np.concatenate([driven] + clamped).astype(np.int64)
    vec = np.zeros(len(pairs))
    vec[:ne] = 1.0
    return pairs, vec, np.unique(driven[:, 0]), np.unique(np.concatenate(clamped)[:, 0])


def quads_target_blocks(geometry: QuadGeometry, target_size: Tuple[int, int], target_shift: Tuple[int, int]):
    """problems/quads_focusing.py:447-451."""
    n1, n2 = geometry.n1_blocks, geometry.n2_blocks
    return np.array([j * n1 + i
                     for i in range((n1 - target_size[0]) // 2 + target_shift[0], (n1 + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2 - target_size[1]) // 2 + target_shift[1], (n2 + target_size[1]) // 2 + target_shift[1])],
                    dtype=np.int32)


def kagome_focusing_constraints(geometry: KagomeGeometry, n_excited_blocks: int, n_blocks_clamped_corner: int = 2):
    """problems/kagome_focusing.py:96-160 (left-loaded only, as in the reference)."""
    n1, n2, ncell = geometry.n1_cells, geometry.n2_cells, geometry.n_cells
    ne, nc = n_excited_blocks, n_blocks_clamped_corner
    blocks = np.arange(2 * n1 * ((n2 - ne) // 2), 2 * n1 * ((n2 + ne) // 2), 2 * n1)
    driven = np.stack([np.tile(blocks, 3), np.array([0] * ne + [1] * ne + [2] * ne)], 1)
    bl = np.concatenate([np.arange(0, nc), [i * n1 for i in range(1, nc)]]) * 2
    br = np.concatenate([np.arange(n1 - nc, n1) * 2, [(i + 1) * 2 * n1 - 1 for i in range(0, nc)]])
    tr = np.concatenate([np.arange(ncell - nc, ncell), [ncell - i * n1 - 1 for i in range(1, nc)]]) * 2 + 1
    tl = np.concatenate([np.arange(ncell - n1, ncell - n1 + nc) * 2 + 1,
                         np.array([ncell - n1 - i * n1 for i in range(0, nc)]) * 2])
    clamped = [_tile3(c) for c in (bl, br, tr, tl)]
    pairs = np.concatenate([driven] + clamped).astype(np.int64)
    vec = np.zeros(len(pairs))
    vec[:ne] = 1.0
    return pairs, vec, np.unique(driven[:, 0]), np.unique(np.concatenate(clamped)[:, 0])


def kagome_target_blocks(geometry: KagomeGeometry, target_size, target_shift):
    """problems/kagome_focusing.py:403-407."""
    n1, n2 = geometry.n1_cells, geometry.n2_cells
    return np.array([(2 * (j * n1 + i), 2 * (j * n1 + i) + 1)
                     for i in range((n1 - target_size[0]) // 2 + target_shift[0], (n1 + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2 - target_size[1]) // 2 + target_shift[1], (n2 + target_size[1]) // 2 + target_shift[1])],
                    dtype=np.int32).flatten()


@dataclass
class QuadsFocusingForward:
    """NumPy counterpart of ``problems/quads_focusing.py:ForwardProblem`` (fields with the same names)."""
    n1_blocks: int
    n2_blocks: int
    spacing: Any
    bond_length: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    amplitude: Any
    loading_rate: Any
    input_delay: Any
    n_excited_blocks: int
    loaded_side: str
    input_shift: int
    simulation_time: Any
    n_timepoints: int
    linearized_strains: bool = False
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    n_blocks_clamped_corner: int = 2
    steps_per_interval: Optional[int] = None
    integrator: str = "dopri5"
    batch: int = 1
    device: int = 0
    name: str = "quads_focusing"
    _lib: Any = None

    def setup(self):
        g = self.geometry = QuadGeometry(self.n1_blocks, self.n2_blocks, self.spacing, self.bond_length)
        self.bond_connectivity = g.bond_connectivity()
        self.reference_bond_vectors = g.reference_bond_vectors()
        pairs, vec, self.driven_blocks_ids, self.clamped_blocks_ids = quads_focusing_constraints(
            g, self.n_excited_blocks, self.loaded_side, self.input_shift, self.n_blocks_clamped_corner)
        self.constrained_block_DOF_pairs = pairs
        self.moving_blocks_ids = np.setdiff1d(np.arange(g.n_blocks), self.clamped_blocks_ids)
        strain = E.build_strain_energy(self.bond_connectivity,
                                       E.ligament_energy_linearized if self.linearized_strains else E.ligament_energy)
        energy = E.combine_block_energies(strain, E.build_contact_energy(self.bond_connectivity)) if self.use_contact else strain
        self.solve_dynamics = setup_dynamic_solver(
            g, energy, constrained_block_DOF_pairs=pairs, constrained_DOFs_fn=L.Pulse(vec),
            damped_blocks=np.arange(g.n_blocks), integrator=self.integrator,
            steps_per_interval=self.steps_per_interval, batch=self.batch, device=self.device, _lib=self._lib)
        self.timepoints = np.linspace(0, self.simulation_time, self.n_timepoints)
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.signed_amplitude = self.amplitude if self.loaded_side in ("left", "bottom") else -self.amplitude
        self.is_setup = True

    def control_params(self, design):
        hs, vs = design
        g = self.geometry
        return ControlParams(
            geometrical_params=GeometricalParams(block_centroids=g.block_centroids(hs, vs),
                                                 centroid_node_vectors=g.centroid_node_vectors(hs, vs)),
            mechanical_params=MechanicalParams(
                bond_params=LigamentParams(self.k_stretch, self.k_shear, self.k_rot, self.reference_bond_vectors),
                density=self.density, damping=self.damping,
                contact_params=ContactParams(min_angle=self.min_angle, cutoff_angle=self.cutoff_angle, k_contact=self.k_contact)),
            constraint_params=dict(amplitude=self.signed_amplitude, loading_rate=self.loading_rate, input_delay=self.input_delay))

    def solve(self, design, keep_trajectory=False):
        """design = (horizontal_shifts, vertical_shifts), or a list of ``batch`` such tuples."""
        many = isinstance(design, list)
        cps = [self.control_params(d) for d in design] if many else self.control_params(design)
        fields = self.solve_dynamics(self.state0, self.timepoints, cps, keep_trajectory=keep_trajectory)
        self._last_design = design
        if many:
            return [SolutionData(cp.geometrical_params.block_centroids, cp.geometrical_params.centroid_node_vectors,
                                 self.bond_connectivity, self.timepoints, f) for cp, f in zip(cps, fields)]
        return SolutionData(cps.geometrical_params.block_centroids, cps.geometrical_params.centroid_node_vectors,
                            self.bond_connectivity, self.timepoints, fields)


class TargetKineticEnergy:
    """objective(design) = sum_t sum_{b in target} m v^2/2 and its gradient w.r.t. the design
    (problems/quads_focusing.py:432-471 + jit(value_and_grad(.)) at :565)."""

    def __init__(self, forward, target_size, target_shift):
        self.forward = forward
        if not getattr(forward, "is_setup", False):
            forward.setup()
        self.target_blocks = quads_target_blocks(forward.geometry, target_size, target_shift)

    def value(self, design):
        sol = self.forward.solve(design)
        sols = sol if isinstance(sol, list) else [sol]
        vals = [E.kinetic_energy(s.fields[:, 1, self.target_blocks, :],
                                 compute_inertia(s.centroid_node_vectors, self.forward.density)[self.target_blocks]) for s in sols]
        return vals if isinstance(sol, list) else vals[0]

    def value_and_grad(self, design):
        fw = self.forward
        fw.solve(design, keep_trajectory=True)
        obj, trees, _ = fw.solve_dynamics.kinetic_energy_value_and_vjp(self.target_blocks)
        many = isinstance(design, list)
        designs = design if many else [design]
        trees = trees if many else [trees]
        grads = [fw.geometry.vjp(d, t.geometrical_params.centroid_node_vectors, t.geometrical_params.block_centroids)
                 for d, t in zip(designs, trees)]
        return (obj, grads) if many else (obj, grads[0])
