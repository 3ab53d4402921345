"""TEST INFRASTRUCTURE -- restatement of the reference's problem layer (the callers of the hot path).

Follows, construct by construct (NumPy standing in for jax.numpy; same tile / arange / concatenate calls, same ordering):

* ``problems/quads_focusing.py:104-222``  driven / clamped block-DOF pairs, loading vector, block id lists, pulse
* ``problems/quads_focusing.py:447-467``  target blocks and the target-kinetic-energy objective
* ``problems/kagome_focusing.py:96-172``  the same for the kagome lattice (left-loaded only), ``:403-424`` targets
* ``problems/quads_focusing_multi_input.py:43-86``  weighted sum over forward problems that share one design
* ``problems/quads_kinetic_energy_static_tuning.py:124-283, 453-478``  static compression + delayed pulse, weighted objective
* ``problems/quads_spin.py:210-222, 391-428`` with ``difflexmm/energy.py:502-519``  harmonic drive, target angular momentum

The objective is evaluated through the oracle's own solver (``oracle.ref_dynamics``) and differentiated with
``torch.autograd`` where the reference calls ``jit(value_and_grad(objective))`` (``problems/quads_focusing.py:565``).
Parity pinning: the reference cannot be imported here (no JAX) and holds no index fixtures, so this file is pinned by the
hand-checkable counts of SURVEY appendix C (42 constrained DOFs on the paper lattice, 48 on the kagome one) and frozen in
``tests/golden/problems_*.npz``; only ``tests/`` imports it.
"""
import math

import numpy as np
import torch

from . import ref_dynamics as OD
from . import ref_energy as OE
from . import ref_geometry as OG

F64 = torch.float64


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float64))


def _dofs012(n):
    return np.array([0] * n + [1] * n + [2] * n)


def quads_constraints(n1_blocks, n2_blocks, n_excited_blocks, loaded_side, input_shift, n_blocks_clamped_corner=2):
    """problems/quads_focusing.py:104-209.  Returns a dict with the arrays the reference builds, under its own names."""
    n_blocks = n1_blocks * n2_blocks
    ne, nc = n_excited_blocks, n_blocks_clamped_corner
    if loaded_side == "left":            # :106-114
        driven = np.array([np.tile(np.arange((n2_blocks - ne) // 2 + input_shift, (n2_blocks + ne) // 2 + input_shift) * n1_blocks, 3),
                           np.array([0] * ne + [1] * ne + [2] * ne)]).T
    elif loaded_side == "right":         # :115-123
        driven = np.array([np.tile(np.arange((n2_blocks - ne) // 2 + input_shift, (n2_blocks + ne) // 2 + input_shift) * n1_blocks
                                   + (n1_blocks - 1), 3),
                           np.array([0] * ne + [1] * ne + [2] * ne)]).T
    elif loaded_side == "bottom":        # :124-132
        driven = np.array([np.tile(np.arange((n1_blocks - ne) // 2 + input_shift, (n1_blocks + ne) // 2 + input_shift), 3),
                           np.array([1] * ne + [0] * ne + [2] * ne)]).T
    elif loaded_side == "top":           # :133-141
        driven = np.array([np.tile(np.arange((n1_blocks - ne) // 2 + input_shift, (n1_blocks + ne) // 2 + input_shift)
                                   + n1_blocks * (n2_blocks - 1), 3),
                           np.array([1] * ne + [0] * ne + [2] * ne)]).T
    else:                                # :142-145
        raise ValueError(f"Unknown loaded_side: {loaded_side}. Should be either 'left', 'right', 'bottom' or 'top'.")
    k = 2 * nc - 1
    bl = np.array([np.tile(np.concatenate([np.arange(0, nc), np.array([0 + i * n1_blocks for i in range(1, nc)], dtype=int)]), 3),
                   _dofs012(k)]).T                                                                            # :147-156
    br = np.array([np.tile(np.concatenate([np.arange(n1_blocks - nc, n1_blocks),
                                           np.array([(i + 1) * n1_blocks - 1 for i in range(1, nc)], dtype=int)]), 3),
                   _dofs012(k)]).T                                                                            # :157-167
    tr = np.array([np.tile(np.concatenate([np.arange(n_blocks - nc, n_blocks),
                                           np.array([n_blocks - i * n1_blocks - 1 for i in range(1, nc)], dtype=int)]), 3),
                   _dofs012(k)]).T                                                                            # :168-177
    tl = np.array([np.tile(np.concatenate([np.arange(n_blocks - n1_blocks, n_blocks - n1_blocks + nc),
                                           np.array([n_blocks - n1_blocks - i * n1_blocks for i in range(1, nc)], dtype=int)]), 3),
                   _dofs012(k)]).T                                                                            # :178-188
    pairs = np.concatenate([driven, bl, br, tr, tl])                                                          # :189-192
    vec = np.zeros((len(pairs),))
    vec[:ne] = 1                                                                                              # :193-196
    clamped_ids = np.unique(np.concatenate([bl, br, tr, tl])[:, 0])                                           # :198-203
    moving_ids = np.setdiff1d(np.arange(n_blocks), clamped_ids)                                               # :204-207
    driven_ids = np.unique(driven[:, 0])                                                                      # :208
    return dict(constrained_block_DOF_pairs=pairs, constrained_DOFs_loading_vector=vec, clamped_blocks_ids=clamped_ids,
                moving_blocks_ids=moving_ids, driven_blocks_ids=driven_ids)


def kagome_constraints(n1_cells, n2_cells, n_excited_blocks, n_blocks_clamped_corner=2, loaded_side="left"):
    """problems/kagome_focusing.py:96-172."""
    n_cells = n1_cells * n2_cells
    ne, nc = n_excited_blocks, n_blocks_clamped_corner
    if loaded_side == "left":            # :99-106
        driven = np.array([np.tile(np.arange(2 * n1_cells * ((n2_cells - ne) // 2), 2 * n1_cells * ((n2_cells + ne) // 2), 2 * n1_cells), 3),
                           np.array([0] * ne + [1] * ne + [2] * ne)]).T
    else:                                # :107-109
        raise ValueError(f"Unknown loaded_side: {loaded_side}. Only 'left' is implemented.")
    bl = np.array([np.tile(np.concatenate([np.arange(0, nc), np.array([0 + i * n1_cells for i in range(1, nc)], dtype=int)]), 3) * 2,
                   _dofs012(2 * nc - 1)]).T                                                                   # :112-121
    br = np.array([np.tile(np.concatenate([np.arange(n1_cells - nc, n1_cells) * 2,
                                           np.array([(i + 1) * 2 * n1_cells - 1 for i in range(0, nc)], dtype=int)]), 3),
                   _dofs012(2 * nc)]).T                                                                       # :122-132
    tr = np.array([np.tile(np.concatenate([np.arange(n_cells - nc, n_cells),
                                           np.array([n_cells - i * n1_cells - 1 for i in range(1, nc)], dtype=int)]) * 2 + 1, 3),
                   _dofs012(2 * nc - 1)]).T                                                                   # :133-141
    tl = np.array([np.tile(np.concatenate([np.arange(n_cells - n1_cells, n_cells - n1_cells + nc) * 2 + 1,
                                           np.array([n_cells - n1_cells - i * n1_cells for i in range(0, nc)], dtype=int) * 2]), 3),
                   _dofs012(2 * nc)]).T                                                                       # :142-152
    pairs = np.concatenate([driven, bl, br, tr, tl])                                                          # :153-156
    vec = np.zeros((len(pairs),))
    vec[:ne] = 1                                                                                              # :157-160
    clamped_ids = np.unique(np.concatenate([bl, br, tr, tl])[:, 0])
    moving_ids = np.setdiff1d(np.arange(2 * n_cells), clamped_ids)
    driven_ids = np.unique(driven[:, 0])
    return dict(constrained_block_DOF_pairs=pairs, constrained_DOFs_loading_vector=vec, clamped_blocks_ids=clamped_ids,
                moving_blocks_ids=moving_ids, driven_blocks_ids=driven_ids)


def pulse(t, amplitude, loading_rate):
    """problems/quads_focusing.py:211-216 (identical in kagome_focusing.py:174-179)."""
    t = _t(t)
    return amplitude * torch.where((t > 0.) & (t < loading_rate ** -1), (1 - torch.cos(2 * math.pi * loading_rate * t)) / 2,
                                   torch.zeros((), dtype=F64))


def harmonic_signal(t, amplitude, loading_rate):
    """problems/quads_spin.py:210-215."""
    t = _t(t)
    return amplitude * torch.where((t > 0.), (1 - torch.cos(2 * math.pi * loading_rate * t)) / 2, torch.zeros((), dtype=F64))


def make_constrained_DOFs_fn(loading_vector, signal=pulse):
    """problems/quads_focusing.py:218-222 (``signal=harmonic_signal``: quads_spin.py:217-222)."""
    vec = _t(loading_vector)

    def constrained_DOFs_fn(t, amplitude, loading_rate, input_delay):
        return signal(_t(t) - input_delay, amplitude, loading_rate) * vec
    return constrained_DOFs_fn


def quads_target_blocks(n1_blocks, n2_blocks, target_size, target_shift):
    """problems/quads_focusing.py:447-451 (and quads_focusing_multi_input.py:58-62)."""
    return np.array([j * n1_blocks + i
                     for i in range((n1_blocks - target_size[0]) // 2 + target_shift[0], (n1_blocks + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2_blocks - target_size[1]) // 2 + target_shift[1], (n2_blocks + target_size[1]) // 2 + target_shift[1])])


def kagome_target_blocks(n1_cells, n2_cells, target_size, target_shift):
    """problems/kagome_focusing.py:403-407."""
    return np.array([(2 * (j * n1_cells + i), 2 * (j * n1_cells + i) + 1)
                     for i in range((n1_cells - target_size[0]) // 2 + target_shift[0], (n1_cells + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2_cells - target_size[1]) // 2 + target_shift[1], (n2_cells + target_size[1]) // 2 + target_shift[1])]).flatten()


class ForwardProblem:
    """``ForwardProblem.setup`` / ``.solve`` of problems/quads_focusing.py:82-317 and kagome_focusing.py:73-273 on the oracle.
    ``lattice``: "quads" (n1 x n2 blocks) or "kagome" (n1 x n2 cells)."""

    def __init__(self, lattice, n1, n2, spacing, bond_length, k_stretch, k_shear, k_rot, density, damping, amplitude, loading_rate,
                 input_delay, n_excited_blocks, simulation_time, n_timepoints, loaded_side="left", input_shift=0,
                 linearized_strains=False, use_contact=True, k_contact=1.0, min_angle=0.0, cutoff_angle=5 * math.pi / 180,
                 n_blocks_clamped_corner=2, signal=pulse):
        self.lattice = lattice
        if lattice == "quads":
            self.geometry = OG.QuadGeometry(n1, n2, spacing, bond_length)
            bc = quads_constraints(n1, n2, n_excited_blocks, loaded_side, input_shift, n_blocks_clamped_corner)
        elif lattice == "rotated_squares":
            # problems/reference_design.py:75-80, 88-197: RotatedSquareGeometry of n1/2 x n2/2 cells, the design is the initial angle;
            # driven / clamped DOF lists are the ones of quads_focusing.py written on the n1 x n2 block grid
            self.geometry = OG.RotatedSquareGeometry(n1 // 2, n2 // 2, spacing, bond_length)
            bc = quads_constraints(n1, n2, n_excited_blocks, loaded_side, input_shift, n_blocks_clamped_corner)
        else:
            basis = spacing * np.array([[1.0, 0.0], [math.cos(math.pi / 3), math.sin(math.pi / 3)]])   # kagome_focusing.py:83-88
            self.geometry = OG.KagomeGeometry(n1, n2, basis, bond_length)
            bc = kagome_constraints(n1, n2, n_excited_blocks, n_blocks_clamped_corner, loaded_side)
        self.__dict__.update(bc)
        self.n1, self.n2 = n1, n2
        self.bonds = self.geometry.bond_connectivity()
        self.reference_bond_vectors = self.geometry.reference_bond_vectors()
        strain = OE.build_strain_energy(self.bonds, OE.ligament_energy_linearized if linearized_strains else OE.ligament_energy)
        energy = OE.combine_block_energies(strain, OE.build_contact_energy(self.bonds)) if use_contact else strain    # :229-237
        self.energy = energy
        self.solver_args = dict(constrained_block_DOF_pairs=self.constrained_block_DOF_pairs,
                                constrained_DOFs_fn=make_constrained_DOFs_fn(self.constrained_DOFs_loading_vector, signal),
                                damped_blocks=np.arange(self.geometry.n_blocks))                               # :101, :239-248
        self.timepoints = np.linspace(0, simulation_time, n_timepoints)                                          # :251
        self.state0 = np.zeros((2, self.geometry.n_blocks, 3))                                                   # :254
        # flip the amplitude if loading from right or top (:257)
        self.signed_amplitude = amplitude if loaded_side in ("left", "bottom") else -amplitude
        self.p = dict(k_stretch=k_stretch, k_shear=k_shear, k_rot=k_rot, density=density, damping=damping, loading_rate=loading_rate,
                      input_delay=input_delay, k_contact=k_contact, min_angle=min_angle, cutoff_angle=cutoff_angle)

    def control_params(self, design):
        """:266-292.  ``design``: tuple of tensors (kept on the autograd tape)."""
        p = self.p
        return OE.ControlParams(
            OE.GeometricalParams(self.geometry.block_centroids(*design), self.geometry.centroid_node_vectors(*design)),
            OE.MechanicalParams(OE.LigamentParams(_t(p["k_stretch"]), _t(p["k_shear"]), _t(p["k_rot"]), _t(self.reference_bond_vectors)),
                                _t(p["density"]), None, _t(p["damping"]),
                                OE.ContactParams(_t(p["min_angle"]), _t(p["cutoff_angle"]), _t(p["k_contact"]))),
            constraint_params=dict(amplitude=_t(self.signed_amplitude), loading_rate=_t(p["loading_rate"]), input_delay=_t(p["input_delay"])))

    def velocity_history(self, design, steps_per_interval):
        """Free-DOF history (T, 2, n_free) of the fixed-grid solve on the autograd tape + the solver (for its DOF ids)."""
        solver = OD.setup_dynamic_solver(self.geometry, self.energy, integrator="fixed", steps_per_interval=steps_per_interval,
                                         **self.solver_args)
        hist, _ = OD.solve_fixed_differentiable(solver, self.geometry, _t(self.state0), self.timepoints, self.control_params(design),
                                                steps_per_interval)
        return hist, solver


def target_kinetic_energy(problem, design, target_blocks, steps_per_interval):
    """problems/quads_focusing.py:453-467: kinetic_energy(fields[:, 1, target_blocks, :], compute_inertia(cnv, density)[target_blocks])
    with energy.py:494-499 (sum over time, blocks and DOFs of m v^2 / 2).  The target blocks are free blocks: their velocities
    are entries of the free-DOF history."""
    hist, solver = problem.velocity_history(design, steps_per_interval)
    free = list(solver.free_DOF_ids)
    cols = torch.as_tensor([[free.index(int(b) * 3 + d) for d in range(3)] for b in target_blocks], dtype=torch.long)
    vel = hist[:, 1][:, cols]                                            # (T, n_target, 3)
    inertia = OG.compute_inertia(problem.geometry.centroid_node_vectors(*design), _t(problem.p["density"]))[torch.as_tensor(target_blocks)]
    return OE.kinetic_energy(vel, inertia)


def split_kinetic_energies(problem, design, target_blocks_list, steps_per_interval):
    """problems/quads_energy_splitting.py:66-83: ONE forward solve, the kinetic energy of every target region (same expression as
    target_kinetic_energy); the objective of :86-87 is ``weights @`` this vector."""
    hist, solver = problem.velocity_history(design, steps_per_interval)
    free = list(solver.free_DOF_ids)
    inertia_all = OG.compute_inertia(problem.geometry.centroid_node_vectors(*design), _t(problem.p["density"]))
    vals = []
    for target_blocks in target_blocks_list:
        cols = torch.as_tensor([[free.index(int(b) * 3 + d) for d in range(3)] for b in target_blocks], dtype=torch.long)
        vals.append(OE.kinetic_energy(hist[:, 1][:, cols], inertia_all[torch.as_tensor(target_blocks)]))
    return torch.stack(vals)


def angular_momentum(block_position, block_velocity, inertia, reference_point):
    """difflexmm/energy.py:502-519."""
    d = block_position[:, :2] - reference_point
    mv = block_velocity[:, :2] * inertia[:, :2]
    momentum_centroids = d[:, 0] * mv[:, 1] - d[:, 1] * mv[:, 0]
    momentum_rotations = block_velocity[:, 2] * inertia[:, 2]
    return momentum_centroids + momentum_rotations


def target_angular_momentum(problem, design, target_blocks, spin_center, steps_per_interval):
    """problems/quads_spin.py:404-428: angular momentum of the target blocks about ``spin_center`` summed over blocks and output
    times; positions = block centroids of the design + displacements, inertia = compute_inertia of the target blocks' vertices."""
    hist, solver = problem.velocity_history(design, steps_per_interval)
    free = list(solver.free_DOF_ids)
    cols = torch.as_tensor([[free.index(int(b) * 3 + d) for d in range(3)] for b in target_blocks], dtype=torch.long)
    tb = torch.as_tensor(np.asarray(target_blocks), dtype=torch.long)
    disp, vel = hist[:, 0][:, cols], hist[:, 1][:, cols]                       # (T, n_target, 3)
    centroids = problem.geometry.block_centroids(*design)[tb]
    inertia = OG.compute_inertia(problem.geometry.centroid_node_vectors(*design)[tb], _t(problem.p["density"]))
    total = torch.zeros((), dtype=F64)
    for k in range(hist.shape[0]):
        total = total + angular_momentum(centroids + disp[k][:, :2], vel[k], inertia, _t(spin_center)).sum()
    return total


def multi_input_objective(problems, design, target_blocks, weights, steps_per_interval):
    """problems/quads_focusing_multi_input.py:64-82: weights @ [target kinetic energy of every forward problem]."""
    vals = torch.stack([target_kinetic_energy(p, design, target_blocks, steps_per_interval) for p in problems])
    return (_t(weights) * vals).sum(), vals


# ---- problems/quads_kinetic_energy_static_tuning.py -------------------------------------------------------------------------
def static_tuning_constraints(n1_blocks, n2_blocks, n_excited_blocks, input_shift):
    """problems/quads_kinetic_energy_static_tuning.py:124-170, construct by construct."""
    n_blocks, ne = n1_blocks * n2_blocks, n_excited_blocks
    driven = np.array([np.tile(np.arange((n2_blocks - ne) // 2 + input_shift, (n2_blocks + ne) // 2 + input_shift) * n1_blocks, 3),
                       np.array([0] * ne + [1] * ne + [2] * ne)]).T                                             # :126-133
    bottom = np.array([np.concatenate([np.arange(0, n1_blocks)] * 3),
                       np.array([1] * n1_blocks + [0] * n1_blocks + [2] * n1_blocks)]).T                        # :135-139
    top = np.array([np.concatenate([np.arange(n_blocks - n1_blocks, n_blocks)] * 3),
                    np.array([1] * n1_blocks + [0] * n1_blocks + [2] * n1_blocks)]).T                           # :140-145
    pairs = np.concatenate([driven, bottom, top])                                                               # :146-149
    vec = np.zeros((len(pairs),))
    dyn = vec.copy()
    dyn[:ne] = 1                                                                                                # :153-154
    sta = vec.copy()
    sta[3 * ne:3 * ne + n1_blocks] = 0.5                                                                        # :155-156
    sta[3 * ne + 3 * n1_blocks:3 * ne + 4 * n1_blocks] = -0.5                                                   # :157-158
    clamped_ids = np.unique(np.concatenate([bottom, top])[:, 0])                                                # :160-164
    moving_ids = np.setdiff1d(np.arange(n_blocks), clamped_ids)
    driven_ids = np.unique(driven[:, 0])
    return dict(constrained_block_DOF_pairs=pairs, constrained_DOFs_loading_vector_dynamic=dyn,
                constrained_DOFs_loading_vector_static=sta, clamped_blocks_ids=clamped_ids, moving_blocks_ids=moving_ids,
                driven_blocks_ids=driven_ids)


def make_static_tuning_fn(n2_blocks, spacing, vec_dynamic, vec_static):
    """problems/quads_kinetic_energy_static_tuning.py:172-196."""
    vd, vs = _t(vec_dynamic), _t(vec_static)

    def constrained_DOFs_fn_dynamic(t, amplitude, loading_rate):                                                # :176-181
        return amplitude * torch.where((t > 0.) & (t < loading_rate ** -1), (1 - torch.cos(2 * math.pi * loading_rate * t)) / 2,
                                       torch.zeros((), dtype=F64)) * vd

    def constrained_DOFs_fn_static(t, compressive_strain, compressive_strain_rate):                             # :187-192
        return (n2_blocks - 1) * spacing * torch.where(t < compressive_strain * compressive_strain_rate ** -1,
                                                       t * compressive_strain_rate, compressive_strain * torch.ones((), dtype=F64)) * vs

    def constrained_DOFs_fn(t, amplitude, loading_rate, compressive_strain, compressive_strain_rate, input_delay):   # :194-195
        t = _t(t)
        amplitude, loading_rate, compressive_strain, compressive_strain_rate, input_delay = (
            _t(amplitude), _t(loading_rate), _t(compressive_strain), _t(compressive_strain_rate), _t(input_delay))
        return constrained_DOFs_fn_static(t, compressive_strain, compressive_strain_rate) + constrained_DOFs_fn_dynamic(
            t - compressive_strain * compressive_strain_rate ** -1 - input_delay, amplitude, loading_rate)
    return constrained_DOFs_fn


class StaticTuningForward:
    """``ForwardProblem.setup`` / ``forward`` of problems/quads_kinetic_energy_static_tuning.py:102-283 on the oracle."""

    def __init__(self, n1, n2, spacing, bond_length, k_stretch, k_shear, k_rot, density, damping, n_excited_blocks, input_shift,
                 simulation_time_dynamic, n_timepoints, linearized_strains=False, use_contact=True, k_contact=1.0, min_angle=0.0,
                 cutoff_angle=5 * math.pi / 180):
        self.geometry = OG.QuadGeometry(n1, n2, spacing, bond_length)
        self.__dict__.update(static_tuning_constraints(n1, n2, n_excited_blocks, input_shift))
        self.n1, self.n2 = n1, n2
        self.bonds = self.geometry.bond_connectivity()
        self.reference_bond_vectors = self.geometry.reference_bond_vectors()
        strain = OE.build_strain_energy(self.bonds, OE.ligament_energy_linearized if linearized_strains else OE.ligament_energy)
        self.energy = OE.combine_block_energies(strain, OE.build_contact_energy(self.bonds)) if use_contact else strain   # :198-205
        self.constrained_DOFs_fn = make_static_tuning_fn(n2, spacing, self.constrained_DOFs_loading_vector_dynamic,
                                                         self.constrained_DOFs_loading_vector_static)
        self.solver_args = dict(constrained_block_DOF_pairs=self.constrained_block_DOF_pairs,
                                constrained_DOFs_fn=self.constrained_DOFs_fn, damped_blocks=np.arange(self.geometry.n_blocks))
        self.simulation_time_dynamic, self.n_timepoints = simulation_time_dynamic, n_timepoints
        self.state0 = np.zeros((2, self.geometry.n_blocks, 3))                                                  # :118
        self.p = dict(k_stretch=k_stretch, k_shear=k_shear, k_rot=k_rot, density=density, damping=damping, k_contact=k_contact,
                      min_angle=min_angle, cutoff_angle=cutoff_angle)

    def control_params(self, design, amplitude, loading_rate, compressive_strain, compressive_strain_rate):
        """:228-256 (``input_delay = 0.1 / loading_rate``, :229)."""
        p = self.p
        input_delay = 0.1 * _t(loading_rate) ** -1
        return OE.ControlParams(
            OE.GeometricalParams(self.geometry.block_centroids(*design), self.geometry.centroid_node_vectors(*design)),
            OE.MechanicalParams(OE.LigamentParams(_t(p["k_stretch"]), _t(p["k_shear"]), _t(p["k_rot"]), _t(self.reference_bond_vectors)),
                                _t(p["density"]), None, _t(p["damping"]),
                                OE.ContactParams(_t(p["min_angle"]), _t(p["cutoff_angle"]), _t(p["k_contact"]))),
            constraint_params=dict(amplitude=_t(amplitude), loading_rate=_t(loading_rate), compressive_strain=_t(compressive_strain),
                                   compressive_strain_rate=_t(compressive_strain_rate), input_delay=input_delay))

    def timepoints(self, loading_rate, compressive_strain, compressive_strain_rate, full_simulation_time=False, n_timepoints=None):
        """:258-272 (plain numbers: the output times are not differentiated)."""
        n = self.n_timepoints if n_timepoints is None else n_timepoints
        t0 = float(compressive_strain) / float(compressive_strain_rate) + 0.1 / float(loading_rate)
        if full_simulation_time:
            return np.linspace(0, self.simulation_time_dynamic + t0, n)
        return np.concatenate([np.array([0.]), np.linspace(t0, t0 + self.simulation_time_dynamic, n)])

    def velocity_history(self, design, row, step_counts, timepoints=None):
        """Free-DOF history of the dynamic-step solve (:280) on a fixed grid, on the autograd tape.  ``row`` = (amplitude,
        loading_rate, compressive_strain, compressive_strain_rate), tensors or numbers."""
        solver = OD.setup_dynamic_solver(self.geometry, self.energy, integrator="fixed", steps_per_interval=step_counts, **self.solver_args)
        ts = self.timepoints(*[float(x.detach()) if isinstance(x, torch.Tensor) else float(x) for x in row[1:]]) if timepoints is None else timepoints
        hist, _ = OD.solve_fixed_differentiable(solver, self.geometry, _t(self.state0), ts, self.control_params(design, *row), step_counts)
        return hist, solver


def static_tuning_objective(problem, design, rows, target_blocks_list, weights, step_counts):
    """problems/quads_kinetic_energy_static_tuning.py:453-478: weights @ [target kinetic energy of every forward input]; the
    fields of the dynamic step drop the first output (``solution[1:]``, :276)."""
    vals = []
    for row, tb in zip(rows, target_blocks_list):
        hist, solver = problem.velocity_history(design, row, step_counts)
        free = list(solver.free_DOF_ids)
        cols = torch.as_tensor([[free.index(int(b) * 3 + d) for d in range(3)] for b in tb], dtype=torch.long)
        vel = hist[1:, 1][:, cols]
        inertia = OG.compute_inertia(problem.geometry.centroid_node_vectors(*design), _t(problem.p["density"]))[torch.as_tensor(tb)]
        vals.append(OE.kinetic_energy(vel, inertia))
    vals = torch.stack(vals)
    return (_t(weights) * vals).sum(), vals


# ---- geometric constraints of the design loop and the restricted design space --------------------------------------------------------

def quads_boundary_nodes(n1_blocks, n2_blocks):
    """problems/quads_focusing.py:477-489: bottom edge (node 3), right edge (node 0), top edge (node 1, from the last block backwards),
    left edge (node 2)."""
    n_blocks = n1_blocks * n2_blocks
    return np.concatenate([
        np.arange(n1_blocks) * 4 + 3,
        np.arange(n1_blocks - 1, n_blocks, n1_blocks) * 4 + 0,
        np.arange(n_blocks - 1, n_blocks - n1_blocks - 1, -1) * 4 + 1,
        np.arange(0, n_blocks, n1_blocks) * 4 + 2,
    ])


def angle_constraints(geometry, design, min_void_angle=0., min_block_angle=0., boundary_angle_constraint=False):
    """problems/quads_focusing.py:473-532 (restricted_space / energy_splitting: the variant without the boundary rows): <= 0 when
    satisfied, concatenation of -(void_1 - min_void), -(void_2 - min_void), -(block_1 - min_block), -(block_2 - min_block), every angle
    taken mod 2 pi, [and -(boundary block angle - min_block)]."""
    node_vectors = geometry.centroid_node_vectors(*design)
    angles = [torch.remainder(a, 2 * math.pi) for a in OG.compute_edge_angles(node_vectors, geometry.bond_connectivity())]
    out = [-(angles[0] - min_void_angle), -(angles[1] - min_void_angle), -(angles[2] - min_block_angle), -(angles[3] - min_block_angle)]
    if boundary_angle_constraint:
        u1, u2 = OG.compute_edge_unit_vectors(node_vectors, quads_boundary_nodes(geometry.n1_blocks, geometry.n2_blocks))
        out.append(-(torch.remainder(OG.angle_between_unit_vectors(u1, u2), 2 * math.pi) - min_block_angle))
    return torch.cat(out)


def edge_length_constraints(geometry, design, min_edge_length):
    """problems/quads_focusing.py:535-544."""
    return -(OG.compute_edge_lengths(geometry.centroid_node_vectors(*design)).reshape(-1) - min_edge_length)


def restricted_space_masks(n1_blocks, n2_blocks, shapes, target_shift, design_patch_size=None):
    """problems/quads_focusing_restricted_space.py:435-455: boolean masks over the horizontal / vertical shift arrays of the window of
    ``design_patch_size`` blocks centred on the target."""
    patch = design_patch_size if design_patch_size is not None else max(n1_blocks, n2_blocks)
    x_start = int(np.clip((n1_blocks - patch) // 2 + target_shift[0], 0, n1_blocks))
    x_end = int(np.clip((n1_blocks + patch) // 2 + target_shift[0], 0, n1_blocks))
    y_start = int(np.clip((n2_blocks - patch) // 2 + target_shift[1], 0, n2_blocks))
    y_end = int(np.clip((n2_blocks + patch) // 2 + target_shift[1], 0, n2_blocks))
    hm = np.full(shapes[0], False)
    hm[x_start:x_end + 1, y_start:y_end] = True
    vm = np.full(shapes[1], False)
    vm[x_start:x_end, y_start:y_end + 1] = True
    return hm, vm


def reduced_to_all_shifts(reduced, initial_guess_all, masks):
    """problems/quads_focusing_restricted_space.py:462-469 (differentiable in the reduced shifts)."""
    out = []
    for r, full, m in zip(reduced, initial_guess_all, masks):
        a = _t(full).clone()
        out.append(a.masked_scatter(torch.as_tensor(m), r))
    return tuple(out)


# ---- hinge characterisation (problems/hinge_characterization.py) -------------------------------------------------------------------------

class HingeForward:
    """``ForwardProblem`` (:18-279) / ``ForwardProblemQuads`` (:281-545) on the oracle: clamped top and bottom rows, the top row ramped to
    ``amplitude`` in tension / compression / shear, damping scaled by (k_stretch, density) of the problem, response = sum of the elastic
    forces dE/du over the driven top-row DOFs at every output time.  ``lattice``: "rotated_squares" (design = (initial_angle,)) or
    "quads" (design = (horizontal_shifts, vertical_shifts))."""

    def __init__(self, lattice, n1, n2, spacing, bond_length, design, k_stretch, density, damping, loading_type, amplitude, loading_rate,
                 n_timepoints, linearized_strains=False, force_multiplier=1.0, use_contact=True, k_contact=1.0, min_angle=0.0,
                 cutoff_angle=5 * math.pi / 180):
        g = self.geometry = OG.RotatedSquareGeometry(n1, n2, spacing, bond_length) if lattice == "rotated_squares" else OG.QuadGeometry(n1, n2, spacing, bond_length)
        design = tuple(_t(d) for d in design)
        self.block_centroids, self.centroid_node_vectors = g.block_centroids(*design), g.centroid_node_vectors(*design)
        self.bonds, self.reference_bond_vectors = g.bond_connectivity(), g.reference_bond_vectors()
        k_ref, mass_ref = k_stretch, density * spacing ** 2                                                         # :85-93
        damping_ref = np.array([(k_ref * mass_ref) ** 0.5, (k_ref * mass_ref) ** 0.5, (k_ref * mass_ref) ** 0.5 * spacing ** 2])
        self.damping_values = damping * damping_ref * np.ones((g.n_blocks, 3))
        constrained_blocks = np.concatenate([np.arange(g.n_blocks - g.n1_blocks, g.n_blocks), np.arange(g.n1_blocks)])   # top row, bottom row
        pairs = np.stack([np.concatenate([constrained_blocks] * 3),
                          np.concatenate([k * np.ones(len(constrained_blocks), dtype=np.int64) for k in (0, 1, 2)])], 1)
        loading_vector = np.zeros(len(pairs))
        if loading_type == "tension":
            top_row = np.where(pairs[:, 1] == 1)[0][:g.n1_blocks]
            loading_vector[top_row] = 1.0
        elif loading_type == "compression":
            top_row = np.where(pairs[:, 1] == 1)[0][:g.n1_blocks]
            loading_vector[top_row] = -1.0
        elif loading_type == "shear":
            top_row = np.where(pairs[:, 1] == 0)[0][:g.n1_blocks]
            loading_vector[top_row] = 1.0
        else:
            raise ValueError("Loading type should be either tension, compression, or shear!")
        self.constrained_block_DOF_pairs, self.loading_vector = pairs, loading_vector
        self.reaction_block_DOF_pairs = pairs[top_row]
        vec = _t(loading_vector)

        def applied_displacement(t, amplitude, loading_rate):                                                        # :134-135
            t = torch.as_tensor(t, dtype=torch.float64)
            return amplitude * torch.where(t < loading_rate ** -1, t * loading_rate, torch.ones_like(t))
        self.applied_displacement = applied_displacement
        strain = OE.build_strain_energy(self.bonds, OE.ligament_energy_linearized if linearized_strains else OE.ligament_energy)
        self.potential_energy = OE.combine_block_energies(strain, OE.build_contact_energy(self.bonds)) if use_contact else strain
        self.solver_args = dict(constrained_block_DOF_pairs=pairs,
                                constrained_DOFs_fn=lambda t, amplitude, loading_rate: vec * applied_displacement(t, amplitude, loading_rate),
                                damped_blocks=np.arange(g.n_blocks))
        self.timepoints = np.linspace(0, loading_rate ** -1, n_timepoints)
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.p = dict(density=density, amplitude=amplitude, loading_rate=loading_rate, k_contact=k_contact, min_angle=min_angle,
                      cutoff_angle=cutoff_angle, force_multiplier=force_multiplier)

    def control_params(self, k_values):
        p = self.p
        return OE.ControlParams(
            OE.GeometricalParams(self.block_centroids, self.centroid_node_vectors),
            OE.MechanicalParams(OE.LigamentParams(k_values[0], k_values[1], k_values[2], _t(self.reference_bond_vectors)), _t(p["density"]),
                                None, _t(self.damping_values), OE.ContactParams(_t(p["min_angle"]), _t(p["cutoff_angle"]), _t(p["k_contact"]))),
            constraint_params=dict(amplitude=_t(p["amplitude"]), loading_rate=_t(p["loading_rate"])))

    def force_displacement(self, k_values, steps_per_interval):
        """:225-244 on the taped fixed-grid solve: (applied displacement (T,), reaction force history (T,) * force_multiplier)."""
        cp = self.control_params(k_values)
        solver = OD.setup_dynamic_solver(self.geometry, self.potential_energy, integrator="fixed", steps_per_interval=steps_per_interval,
                                         **self.solver_args)
        hist, _ = OD.solve_fixed_differentiable(solver, self.geometry, _t(self.state0), self.timepoints, cp, steps_per_interval)
        rb, rd = self.reaction_block_DOF_pairs[:, 0], self.reaction_block_DOF_pairs[:, 1]
        forces = []
        for k, t in enumerate(self.timepoints):
            u = solver.kinematics(hist[k, 0], float(t), cp.constraint_params)                      # all DOFs, prescribed ones included
            if not u.requires_grad:
                u = u.clone().requires_grad_(True)
            f = torch.autograd.grad(self.potential_energy(u, cp), u, create_graph=True)[0]       # elastic_forces = grad(potential_energy)
            forces.append(f[rb, rd].sum())
        applied = self.applied_displacement(self.timepoints, cp.constraint_params["amplitude"], cp.constraint_params["loading_rate"])
        return applied, torch.stack(forces) * self.p["force_multiplier"]


def hinge_response_squared_error(problems, target_forces, k_values, steps_per_interval):
    """:637-646: mean over tests and output times of (reaction force - target force)^2."""
    forces = torch.stack([p.force_displacement(k_values, steps_per_interval)[1] for p in problems])
    return ((forces - _t(target_forces)) ** 2).mean()
