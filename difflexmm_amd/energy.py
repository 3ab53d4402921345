"""Energy builders with the reference's names (``difflexmm/energy.py``).  Each returns a callable that
(a) carries the declarative ``spec`` the HIP engine consumes -- bond model, contact on/off, connectivity --
and (b) can still be called as ``energy_fn(block_displacement, control_params)`` on the host (NumPy) for
post-processing and tests.  The per-time-step evaluation never goes through these Python bodies.
"""
import numpy as np

from ._binding import BOND_LINEARIZED, BOND_NONLINEAR, BOND_SIMPLE_SPRING, BOND_STRETCH_TORSION
from .geometry import compute_edge_angles, rotation_matrix


def block_to_node_kinematics(block_displacement, centroid_node_vectors):
    """kinematics.py:13-37: (n_blocks,3),(n_blocks,n_npb,2) -> (n_blocks,n_npb,3)."""
    u = np.asarray(block_displacement, dtype=float)
    r = np.asarray(centroid_node_vectors, dtype=float)
    c, s = np.cos(u[:, 2]), np.sin(u[:, 2])
    R = np.stack([np.stack([c - 1, -s], -1), np.stack([s, c - 1], -1)], -2)
    xy = u[:, None, :2] + np.einsum("bij,bnj->bni", R, r)
    return np.concatenate([xy, np.broadcast_to(u[:, None, 2:3], xy.shape[:2] + (1,))], -1)


def _block_to_node_displacement(block_displacement, centroid_node_vector):
    """kinematics.py:13-31 for one block / one node."""
    return block_to_node_kinematics(np.asarray(block_displacement)[None], np.asarray(centroid_node_vector)[None, None])[0, 0]


def ligament_strains_linearized(DOFs1, DOFs2, reference_vector=np.array([1., 0.])):
    """energy.py:70-96."""
    DOFs1, DOFs2, rv = np.asarray(DOFs1), np.asarray(DOFs2), np.asarray(reference_vector, dtype=float)
    dU = DOFs2[:, :2] - DOFs1[:, :2]
    l2 = np.sum(rv * rv, -1)
    axial = np.sum(dU * rv, -1) / l2
    shear = (rv[..., 0] * dU[:, 1] - rv[..., 1] * dU[:, 0]) / l2 - (DOFs2[:, 2] + DOFs1[:, 2]) / 2
    return axial, shear, DOFs2[:, 2] - DOFs1[:, 2]


def ligament_strains(DOFs1, DOFs2, reference_vector=np.array([1., 0.])):
    """energy.py:120-155."""
    DOFs1, DOFs2 = np.asarray(DOFs1), np.asarray(DOFs2)
    rv = np.broadcast_to(np.asarray(reference_vector, dtype=float), DOFs1[:, :2].shape)
    cur = DOFs2[:, :2] - DOFs1[:, :2] + rv
    mean = (DOFs2[:, 2] + DOFs1[:, 2]) / 2
    c, s = np.cos(mean), np.sin(mean)
    pushed = np.stack([c * rv[:, 0] - s * rv[:, 1], s * rv[:, 0] + c * rv[:, 1]], -1)
    axial = np.sqrt(np.sum(cur * cur, -1) / np.sum(rv * rv, -1)) - 1
    shear = np.mod(np.arctan2(cur[:, 1], cur[:, 0]) - np.arctan2(pushed[:, 1], pushed[:, 0]) + np.pi, 2 * np.pi) - np.pi
    return axial, shear, DOFs2[:, 2] - DOFs1[:, 2]


class _BondEnergy:
    def __init__(self, name, model, strains):
        self.__name__, self.model, self._strains = name, model, strains

    def __call__(self, nodal_DOFs, reference_vector=np.array([1., 0.]), k_stretch=1., k_shear=1., k_rot=1.):
        axial, shear, drot = self._strains(*nodal_DOFs, reference_vector=reference_vector)
        l0 = np.linalg.norm(np.asarray(reference_vector, dtype=float), axis=-1)
        return k_stretch * (axial * l0) ** 2 / 2 + k_shear * (shear * l0) ** 2 / 2 + k_rot * drot ** 2 / 2


#: energy.py:99-117
ligament_energy_linearized = _BondEnergy("ligament_energy_linearized", BOND_LINEARIZED, ligament_strains_linearized)
#: energy.py:158-176
ligament_energy = _BondEnergy("ligament_energy", BOND_NONLINEAR, ligament_strains)


class _SimpleSpring(_BondEnergy):
    """energy.py:30-48."""

    def __init__(self):
        self.__name__, self.model = "simple_spring_energy", BOND_SIMPLE_SPRING

    def __call__(self, nodal_DOFs, reference_vector=np.array([1., 0.]), k_stretch=1., **_ignored):
        D1, D2 = (np.asarray(a, dtype=float) for a in nodal_DOFs)
        rv = np.asarray(reference_vector, dtype=float)
        l = np.linalg.norm(D2[:, :2] - D1[:, :2] + rv, axis=-1)
        l0 = np.linalg.norm(rv, axis=-1)
        return k_stretch * ((l / l0 - 1) * l0) ** 2 / 2


class _StretchTorsionSpring(_BondEnergy):
    """energy.py:51-67."""

    def __init__(self):
        self.__name__, self.model = "stretching_torsional_spring_energy", BOND_STRETCH_TORSION

    def __call__(self, nodal_DOFs, k_stretch=1., k_rot=1., **_ignored):
        D1, D2 = (np.asarray(a, dtype=float) for a in nodal_DOFs)
        dU, dRot = D2[:, :2] - D1[:, :2], D2[:, 2] - D1[:, 2]
        return k_stretch * np.sum(dU * dU, -1) / 2 + k_rot * dRot ** 2 / 2


#: energy.py:30-48 -- linear spring between two nodes, rest length |reference_vector| (bond_params: anything with k_stretch and
#: reference_vector, e.g. LigamentParams; its k_shear / k_rot are not used)
simple_spring_energy = _SimpleSpring()
#: energy.py:51-67 -- zero-length spring with a torsional term (bond_params: StretchingTorsionalSpringParams)
stretching_torsional_spring_energy = _StretchTorsionSpring()


class EnergySpec:
    """What the engine needs to know about a potential energy: connectivity, bond model, contact."""

    def __init__(self, bond_connectivity=None, bond_model=None, contact=False):
        self.bond_connectivity = None if bond_connectivity is None else np.asarray(bond_connectivity, dtype=np.int32).reshape(-1, 2)
        self.bond_model = bond_model
        self.contact = contact

    def merge(self, other):
        if self.bond_connectivity is not None and other.bond_connectivity is not None and \
                not np.array_equal(self.bond_connectivity, other.bond_connectivity):
            raise ValueError("combined energies must share one bond_connectivity")
        if self.bond_model is not None and other.bond_model is not None:
            raise ValueError("only one strain energy may be combined")
        return EnergySpec(self.bond_connectivity if self.bond_connectivity is not None else other.bond_connectivity,
                          self.bond_model if self.bond_model is not None else other.bond_model,
                          max(int(self.contact or 0), int(other.contact or 0)))


class _EnergyFn:
    def __init__(self, spec, host_fn):
        self.spec, self._host_fn = spec, host_fn

    def __call__(self, block_displacement, control_params):
        return self._host_fn(block_displacement, control_params)


def strain_energy_bond(bond_connectivity, bond_energy_fn=ligament_energy_linearized):
    """energy.py:179-197 (+ jax_md.smap.bond): total energy of a set of bonds from nodal DOFs."""
    bonds = np.asarray(bond_connectivity, dtype=np.int64).reshape(-1, 2)

    def fn(node_displacements, **kwargs):
        nd = np.asarray(node_displacements)
        return np.sum(bond_energy_fn((nd[bonds[:, 0]], nd[bonds[:, 1]]), **kwargs))
    return fn


def build_strain_energy(bond_connectivity, bond_energy_fn=ligament_energy_linearized):
    """energy.py:410-449."""
    if not isinstance(bond_energy_fn, _BondEnergy):
        raise TypeError("bond_energy_fn must be one of difflexmm_amd.energy.ligament_energy, ligament_energy_linearized, "
                        "simple_spring_energy, stretching_torsional_spring_energy")
    bonds_fn = strain_energy_bond(bond_connectivity, bond_energy_fn)

    def host(block_displacement, control_params):
        cnv = control_params.geometrical_params.centroid_node_vectors
        nd = block_to_node_kinematics(block_displacement, cnv).reshape(-1, 3)
        return bonds_fn(nd, **control_params.mechanical_params.bond_params._asdict())
    return _EnergyFn(EnergySpec(bond_connectivity, bond_energy_fn.model, False), host)


def void_angles(current_block_nodes, bond_connectivity):
    """energy.py:204-219."""
    a1, a2, _, _ = compute_edge_angles(current_block_nodes, bond_connectivity)
    return np.concatenate([a1, a2])


def contact_energy(current_void_angles, min_angle=0., cutoff_angle=2.0 * np.pi / 180, k_contact=1.0):
    """energy.py:333-361."""
    a = np.asarray(current_void_angles, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (a - cutoff_angle) / (cutoff_angle - min_angle)
        inner = k_contact / 4 * (cutoff_angle - min_angle) ** 2 * (1 / (x + 1) - 1 / (x - 1) - 2)
    return np.where(a < min_angle, 0.0, np.where(a < cutoff_angle, inner, 0.0))


def point_to_edge_distance(point, edge):
    """energy.py:222-251 (one point (2,), one edge (2, 2))."""
    x0, x1 = np.asarray(edge[0], dtype=float), np.asarray(edge[1], dtype=float)
    p = np.asarray(point, dtype=float)
    t = np.dot(p - x0, x1 - x0) / np.dot(x1 - x0, x1 - x0)
    if 0 <= t <= 1:
        return np.sum((p - x0) ** 2 - (t * (x1 - x0)) ** 2) ** 0.5
    return np.sum((p - x0) ** 2) ** 0.5 if t < 0 else np.sum((p - x1) ** 2) ** 0.5


def edges_distance(edge_1, edge_2):
    """energy.py:254-276."""
    return min([point_to_edge_distance(q, edge_1) for q in edge_2] + [point_to_edge_distance(q, edge_2) for q in edge_1])


def build_void_edge_distance(bond_connectivity):
    """energy.py:283-330: (2 n_bonds,) distances between the block edges on either side of every ligament."""
    bonds = np.asarray(bond_connectivity, dtype=np.int64).reshape(-1, 2)

    def void_edge_distance(current_block_nodes):
        nodes = np.asarray(current_block_nodes, dtype=float)
        n = nodes.shape[1]
        b1, k1, b2, k2 = bonds[:, 0] // n, bonds[:, 0] % n, bonds[:, 1] // n, bonds[:, 1] % n
        p1, p1p, p1n = nodes[b1, k1], nodes[b1, (k1 - 1) % n], nodes[b1, (k1 + 1) % n]
        p2, p2p, p2n = nodes[b2, k2], nodes[b2, (k2 - 1) % n], nodes[b2, (k2 + 1) % n]
        d1 = [edges_distance((a, an), (b, bp)) for a, an, b, bp in zip(p1, p1n, p2, p2p)]
        d2 = [edges_distance((a, ap), (b, bn)) for a, ap, b, bn in zip(p1, p1p, p2, p2n)]
        return np.array(d1 + d2)
    return void_edge_distance


def build_contact_energy(bond_connectivity, angle_based=True):
    """energy.py:364-407: angle-based (default) or distance-based (``angle_based=False``: the penalty of energy.py:333-361 on the
    void-edge distances of energy.py:283-330, ``contact_params`` then being lengths, utils.py:101)."""
    distance_fn = (lambda nodes: void_angles(nodes, bond_connectivity)) if angle_based else build_void_edge_distance(bond_connectivity)

    def host(block_displacement, control_params):
        gp = control_params.geometrical_params
        nd = block_to_node_kinematics(block_displacement, gp.centroid_node_vectors)[:, :, :2]
        nodes = np.asarray(gp.block_centroids)[:, None] + np.asarray(gp.centroid_node_vectors) + nd
        cp = control_params.mechanical_params.contact_params
        return np.sum(contact_energy(distance_fn(nodes), **cp._asdict()))
    return _EnergyFn(EnergySpec(bond_connectivity, None, 1 if angle_based else 2), host)


def combine_block_energies(*energy_fns):
    """energy.py:452-470."""
    spec = EnergySpec()
    for fn in energy_fns:
        if not isinstance(fn, _EnergyFn):
            raise TypeError("combine_block_energies: every energy must come from build_strain_energy / build_contact_energy")
        spec = spec.merge(fn.spec)
    return _EnergyFn(spec, lambda u, cp: sum(fn(u, cp) for fn in energy_fns))


def kinetic_energy(block_velocity, inertia):
    """energy.py:494-499."""
    return np.sum(np.asarray(inertia) * np.asarray(block_velocity) ** 2 / 2)


def angular_momentum(block_position, block_velocity, inertia, reference_point=np.array([0., 0.])):
    """energy.py:502-519."""
    p = np.asarray(block_position)[:, :2] - reference_point
    mv = np.asarray(block_velocity)[:, :2] * np.asarray(inertia)[:, :2]
    return p[:, 0] * mv[:, 1] - p[:, 1] * mv[:, 0] + np.asarray(block_velocity)[:, 2] * np.asarray(inertia)[:, 2]


def compute_ligament_strains(block_displacement, centroid_node_vectors, bond_connectivity, reference_bond_vectors):
    """energy.py:522-529."""
    nd = block_to_node_kinematics(block_displacement, centroid_node_vectors).reshape(-1, 3)
    b = np.asarray(bond_connectivity)
    return ligament_strains(nd[b[:, 0]], nd[b[:, 1]], reference_vector=reference_bond_vectors)


def compute_ligament_strains_history(block_displacement_history, centroid_node_vectors, bond_connectivity, reference_bond_vectors):
    """energy.py:532-534."""
    out = [compute_ligament_strains(u, centroid_node_vectors, bond_connectivity, reference_bond_vectors)
           for u in block_displacement_history]
    return tuple(np.stack([o[i] for o in out]) for i in range(3))
