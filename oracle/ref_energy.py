"""Oracle restatement of ``difflexmm/energy.py`` + ``difflexmm/kinematics.py`` (test infrastructure).

torch-fp64; forces are obtained with ``torch.autograd`` exactly where the reference uses
``jax.grad`` (``jax_md.quantity.force``, ``dynamics.py:31``).
"""
import math
from typing import Any, NamedTuple, Optional

import numpy as np
import torch

# the oracle's tensors are tiny (a lattice's blocks): the intra-op thread pool only adds hand-over time (KATs 13 % faster with one thread)
torch.set_num_threads(1)

from .ref_geometry import F64, _t, compute_edge_angles, rotation_matrix


# --- parameter trees, utils.py:48-163 (field-for-field) --------------------------------------

class GeometricalParams(NamedTuple):
    block_centroids: Any
    centroid_node_vectors: Any


class LigamentParams(NamedTuple):
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    reference_vector: Any


class ContactParams(NamedTuple):
    min_angle: Any
    cutoff_angle: Any
    k_contact: Any


class MechanicalParams(NamedTuple):
    bond_params: Any
    density: Any
    inertia: Optional[Any] = None
    damping: Any = 0.0
    contact_params: Optional[ContactParams] = None


class ControlParams(NamedTuple):
    geometrical_params: GeometricalParams
    mechanical_params: MechanicalParams
    magnetic_params: Optional[Any] = None
    loading_params: dict = {}
    constraint_params: dict = {}


# --- kinematics.py:13-37 ---------------------------------------------------------------------

def block_to_node_kinematics(block_displacement, centroid_node_vectors):
    """kinematics.py:13-37: U_n = u_xy + (R(theta) - I) r_n ; third component = theta.
    (n_blocks,3), (n_blocks,n_npb,2) -> (n_blocks,n_npb,3)."""
    u = block_displacement
    R = rotation_matrix(u[:, 2]) - torch.eye(2, dtype=F64)
    xy = u[:, None, :2] + torch.einsum("bij,bnj->bni", R, centroid_node_vectors)
    th = u[:, None, 2:3].expand(-1, centroid_node_vectors.shape[1], 1)
    return torch.cat([xy, th], -1)


# --- bond energies, energy.py:70-176 -----------------------------------------------------------

def _norm(v):
    return torch.linalg.norm(v, dim=-1)


def ligament_strains_linearized(D1, D2, reference_vector):
    """energy.py:70-96."""
    dU = D2[:, :2] - D1[:, :2]
    dRot = D2[:, 2] - D1[:, 2]
    l2 = _norm(reference_vector) ** 2
    axial = torch.sum(dU * reference_vector, -1) / l2
    cross = reference_vector[..., 0] * dU[:, 1] - reference_vector[..., 1] * dU[:, 0]
    shear = cross / l2 - (D2[:, 2] + D1[:, 2]) / 2
    return axial, shear, dRot


def ligament_strains(D1, D2, reference_vector):
    """energy.py:120-155 (nonlinear strains; shear wrapped into [-pi, pi) by a floor-mod)."""
    dU = D2[:, :2] - D1[:, :2]
    dRot = D2[:, 2] - D1[:, 2]
    mean_rot = (D2[:, 2] + D1[:, 2]) / 2
    cur = dU + reference_vector
    cur_angle = torch.atan2(cur[:, 1], cur[:, 0])                      # energy.py:143-144
    ref = torch.ones((len(D1), 2), dtype=F64) * reference_vector
    pushed = torch.einsum("bij,bj->bi", rotation_matrix(mean_rot), ref)  # energy.py:145-146
    pushed_angle = torch.atan2(pushed[:, 1], pushed[:, 0])
    axial = (torch.sum(cur * cur, -1) / torch.sum(reference_vector * reference_vector, -1)) ** 0.5 - 1
    shear = torch.remainder(cur_angle - pushed_angle + math.pi, 2 * math.pi) - math.pi  # jnp.mod == floor-mod
    return axial, shear, dRot


def _bond_energy(strains, reference_vector, k_stretch, k_shear, k_rot):
    axial, shear, dRot = strains
    l0 = _norm(reference_vector)
    return k_stretch * (axial * l0) ** 2 / 2 + k_shear * (shear * l0) ** 2 / 2 + k_rot * dRot ** 2 / 2


def ligament_energy_linearized(nodal_DOFs, reference_vector, k_stretch=1.0, k_shear=1.0, k_rot=1.0):
    """energy.py:99-117 (per-bond energies, not yet summed)."""
    return _bond_energy(ligament_strains_linearized(*nodal_DOFs, reference_vector), reference_vector, k_stretch, k_shear, k_rot)


def ligament_energy(nodal_DOFs, reference_vector, k_stretch=1.0, k_shear=1.0, k_rot=1.0):
    """energy.py:158-176."""
    return _bond_energy(ligament_strains(*nodal_DOFs, reference_vector), reference_vector, k_stretch, k_shear, k_rot)


class StretchingTorsionalSpringParams(NamedTuple):
    """utils.py:80-91."""
    k_stretch: object
    k_rot: object


class SimpleSpringParams(NamedTuple):
    """The keyword arguments of simple_spring_energy (energy.py:30): the reference has no NamedTuple for them."""
    k_stretch: object
    reference_vector: object


def simple_spring_energy(nodal_DOFs, reference_vector, k_stretch=1.0):
    """energy.py:30-48."""
    D1, D2 = nodal_DOFs
    dU = D2[:, :2] - D1[:, :2]
    l = _norm(dU + reference_vector)
    l0 = _norm(reference_vector)
    axial_strain = l / l0 - 1
    return k_stretch * (axial_strain * l0) ** 2 / 2


def stretching_torsional_spring_energy(nodal_DOFs, k_stretch=1.0, k_rot=1.0):
    """energy.py:51-67."""
    D1, D2 = nodal_DOFs
    dU = D2[:, :2] - D1[:, :2]
    dRot = D2[:, 2] - D1[:, 2]
    return k_stretch * torch.sum(dU * dU, -1) / 2 + k_rot * dRot ** 2 / 2


def strain_energy_bond(bond_connectivity, bond_energy_fn=ligament_energy_linearized):
    """energy.py:179-197 + jax_md.smap.bond (0.2.5): gather both end nodes, per-bond energy, sum."""
    bonds = torch.as_tensor(np.asarray(bond_connectivity), dtype=torch.long)

    def fn(node_displacements, **kwargs):
        Ra, Rb = node_displacements[bonds[:, 0]], node_displacements[bonds[:, 1]]
        return torch.sum(bond_energy_fn((Ra, Rb), **kwargs))

    return fn


def build_strain_energy(bond_connectivity, bond_energy_fn=ligament_energy_linearized):
    """energy.py:410-449."""
    bonds_fn = strain_energy_bond(bond_connectivity, bond_energy_fn)

    def strain_energy_fn(block_displacement, control_params):
        cnv = control_params.geometrical_params.centroid_node_vectors
        bp = control_params.mechanical_params.bond_params
        nd = block_to_node_kinematics(block_displacement, cnv).reshape(-1, 3)
        return bonds_fn(nd, **{k: _t(v) for k, v in bp._asdict().items()})

    return strain_energy_fn


# --- contact, energy.py:204-219, 333-407 ---------------------------------------------------------

def void_angles(current_block_nodes, bond_connectivity):
    """energy.py:204-219: the two void angles per bond, all first angles then all second angles."""
    a1, a2, _, _ = compute_edge_angles(current_block_nodes, bond_connectivity)
    return torch.cat([a1, a2])


def contact_energy(current_void_angles, min_angle=0.0, cutoff_angle=2.0 * math.pi / 180, k_contact=1.0):
    """energy.py:333-361 (nested where; masked branch still evaluated, as in the reference)."""
    min_angle, cutoff_angle, k_contact = _t(min_angle), _t(cutoff_angle), _t(k_contact)
    a = current_void_angles
    x = (a - cutoff_angle) / (cutoff_angle - min_angle)
    inner = k_contact / 4 * (cutoff_angle - min_angle) ** 2 * ((x + 1) ** -1 - (x - 1) ** -1 - 2)
    zero = torch.zeros_like(a)
    return torch.where(a < min_angle, zero, torch.where(a < cutoff_angle, inner, zero))


def point_to_edge_distance(point, edge):
    """energy.py:222-251, batched over a leading axis: point (n, 2), edge (n, 2, 2).  The reference's nested ``jnp.where``
    evaluates all three branches; here the value of every branch is the reference's formula and the branch is picked by the
    same conditions -- the unselected square roots are guarded (the usual double-where) so that autograd does not turn an
    unselected sqrt(0) into NaN * 0."""
    x0, x1 = edge[:, 0], edge[:, 1]
    t = torch.sum((point - x0) * (x1 - x0), -1) / torch.sum((x1 - x0) * (x1 - x0), -1)
    inside = (t >= 0) & (t <= 1)
    arg_in = torch.sum((point - x0) ** 2 - (t[:, None] * (x1 - x0)) ** 2, -1)
    d_in = torch.where(inside, arg_in, torch.ones_like(arg_in)) ** 0.5
    arg0 = torch.sum((point - x0) ** 2, -1)
    d0 = torch.where(t < 0, arg0, torch.ones_like(arg0)) ** 0.5
    arg1 = torch.sum((point - x1) ** 2, -1)
    d1 = torch.where(~inside & ~(t < 0), arg1, torch.ones_like(arg1)) ** 0.5
    return torch.where(inside, d_in, torch.where(t < 0, d0, d1))


def edges_distance_mapped(edge_1, edge_2):
    """energy.py:254-280: min over the four point-to-other-edge distances, edges (n, 2, 2)."""
    e2_onto_e1 = torch.stack([point_to_edge_distance(edge_2[:, k], edge_1) for k in range(2)], -1)
    e1_onto_e2 = torch.stack([point_to_edge_distance(edge_1[:, k], edge_2) for k in range(2)], -1)
    return torch.min(torch.cat([e2_onto_e1, e1_onto_e2], -1), dim=-1).values


def build_void_edge_distance(bond_connectivity):
    """energy.py:283-330."""
    bonds = torch.as_tensor(np.asarray(bond_connectivity), dtype=torch.long)

    def void_edge_distance(current_block_nodes):
        n = current_block_nodes.shape[1]
        n1, n2 = bonds[:, 0], bonds[:, 1]
        pts1 = current_block_nodes[n1 // n, n1 % n]
        pts1_prev = current_block_nodes[n1 // n, (n1 - 1) % n]
        pts1_next = current_block_nodes[n1 // n, (n1 + 1) % n]
        pts2 = current_block_nodes[n2 // n, n2 % n]
        pts2_prev = current_block_nodes[n2 // n, (n2 - 1) % n]
        pts2_next = current_block_nodes[n2 // n, (n2 + 1) % n]
        d1 = edges_distance_mapped(torch.stack([pts1, pts1_next], 1), torch.stack([pts2, pts2_prev], 1))
        d2 = edges_distance_mapped(torch.stack([pts1, pts1_prev], 1), torch.stack([pts2, pts2_next], 1))
        return torch.cat([d1, d2])

    return void_edge_distance


def build_contact_energy(bond_connectivity, angle_based=True):
    """energy.py:364-407: the penalty of energy.py:333-361 on the void angles (default) or on the void-edge distances."""
    void_edge_distance_fn = build_void_edge_distance(bond_connectivity)
    distance_fn = (lambda nodes: void_angles(nodes, bond_connectivity)) if angle_based else void_edge_distance_fn

    def contact_energy_fn(block_displacement, control_params):
        c = control_params.geometrical_params.block_centroids
        cnv = control_params.geometrical_params.centroid_node_vectors
        cp = control_params.mechanical_params.contact_params
        nd = block_to_node_kinematics(block_displacement, cnv)[:, :, :2]
        nodes = c[:, None] + cnv + nd
        return torch.sum(contact_energy(distance_fn(nodes), **cp._asdict()))

    return contact_energy_fn


def combine_block_energies(*energy_fns):
    """energy.py:452-470."""
    def combined(block_displacement, control_params):
        e = torch.zeros((), dtype=F64)
        for fn in energy_fns:
            e = e + fn(block_displacement, control_params)
        return e
    return combined


def kinetic_energy(block_velocity, inertia):
    """energy.py:494-499."""
    return torch.sum(inertia * block_velocity ** 2 / 2)
