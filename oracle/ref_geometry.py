"""Oracle restatement of ``difflexmm/geometry.py`` on torch-fp64 tensors (test infrastructure).

Every function cites the reference lines it follows.  Tensors are float64; all maps are
differentiable with ``torch.autograd`` so the oracle can produce design gradients the same
way the reference does with ``jax.grad``.
"""
import math

import numpy as np
import torch

F64 = torch.float64


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=F64)


def rotation_matrix(angle):
    """geometry.py:17-23 -- [[c,-s],[s,c]]; ``angle`` may be batched (..., ) -> (..., 2, 2)."""
    angle = _t(angle)
    c, s = torch.cos(angle), torch.sin(angle)
    return torch.stack([torch.stack([c, -s], -1), torch.stack([s, c], -1)], -2)


# --- polygon properties, geometry.py:71-160 -------------------------------------------------

def polygon_area(v):
    """geometry.py:71-84 (shoelace, abs). v: (..., n, 2)."""
    v1 = torch.roll(v, 1, dims=-2)
    return torch.abs(torch.sum(v1[..., 0] * v[..., 1] - v1[..., 1] * v[..., 0], -1) / 2)


def polygon_centroid(v):
    """geometry.py:87-106."""
    area = polygon_area(v)
    v1 = torch.roll(v, 1, dims=-2)
    cr = v1[..., 0] * v[..., 1] - v1[..., 1] * v[..., 0]
    s = v1 + v
    return torch.stack([torch.sum(s[..., 0] * cr, -1), torch.sum(s[..., 1] * cr, -1)], -1) / (6 * area)[..., None]


def polygon_polar_moment(v):
    """geometry.py:109-127 (about the centroid)."""
    c = polygon_centroid(v)[..., None, :]
    a = torch.roll(v, 1, dims=-2) - c
    b = v - c
    cr = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    q = a[..., 0] ** 2 + a[..., 0] * b[..., 0] + b[..., 0] ** 2 + a[..., 1] ** 2 + a[..., 1] * b[..., 1] + b[..., 1] ** 2
    return torch.abs(torch.sum(cr * q, -1) / 12)


def compute_inertia(vertices, density):
    """geometry.py:144-160 -> (n_blocks, 3) = [rho A, rho A, rho I_p]."""
    vertices = _t(vertices)
    area = polygon_area(vertices)
    ip = polygon_polar_moment(vertices)
    density = _t(density)
    m = density * area
    return torch.stack([m, m, density * ip], -1)


def DOFsInfo(n_blocks, constrained_block_DOF_pairs):
    """geometry.py:163-178 (same ordering: constrained ids keep the caller's order, free ids ascending)."""
    pairs = np.asarray(constrained_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
    constrained = pairs[:, 0] * 3 + pairs[:, 1]
    all_ids = np.arange(n_blocks * 3)
    mask = np.ones(n_blocks * 3, dtype=bool)
    mask[constrained] = False
    return all_ids[mask], constrained, all_ids


# --- edge angles, geometry.py:181-253 ---------------------------------------------------------

def compute_edge_unit_vectors(nodes, node_id):
    """geometry.py:181-202, vectorised over an int array ``node_id``. nodes: (n_blocks, n_npb, 2)."""
    n_sides = nodes.shape[1]
    node_id = torch.as_tensor(node_id, dtype=torch.long)
    b, l = node_id // n_sides, node_id % n_sides
    p = nodes[b, l]
    u1 = nodes[b, (l + 1) % n_sides] - p
    u1 = u1 / torch.linalg.norm(u1, dim=-1, keepdim=True)
    u2 = nodes[b, (l - 1) % n_sides] - p
    u2 = u2 / torch.linalg.norm(u2, dim=-1, keepdim=True)
    return u1, u2


def angle_between_unit_vectors(u1, u2):
    """geometry.py:221-231."""
    return torch.atan2(u1[..., 0] * u2[..., 1] - u1[..., 1] * u2[..., 0],
                       u1[..., 0] * u2[..., 0] + u1[..., 1] * u2[..., 1])


def compute_edge_angles(nodes, bonds):
    """geometry.py:234-253, vectorised over bonds (n_bonds, 2) -> 4 arrays (n_bonds,)."""
    bonds = np.asarray(bonds)
    b1n1, b1n2 = compute_edge_unit_vectors(nodes, bonds[:, 0])
    b2n1, b2n2 = compute_edge_unit_vectors(nodes, bonds[:, 1])
    return (angle_between_unit_vectors(b2n2, b1n1), angle_between_unit_vectors(b1n2, b2n1),
            angle_between_unit_vectors(b1n1, b1n2), angle_between_unit_vectors(b2n1, b2n2))


def compute_edge_lengths(cnv):
    """geometry.py:205-218."""
    return torch.linalg.norm(torch.roll(cnv, 1, dims=1) - cnv, dim=2)


# --- lattices ---------------------------------------------------------------------------------

class RotatedSquareGeometry:
    """geometry.py:354-443."""

    def __init__(self, n1_cells, n2_cells, spacing=1.0, bond_length=0.1):
        self.n1_cells, self.n2_cells = n1_cells, n2_cells
        self.spacing, self.bond_length = spacing, bond_length
        self.n1_blocks, self.n2_blocks = 2 * n1_cells, 2 * n2_cells
        self.n_blocks = self.n1_blocks * self.n2_blocks
        self.n_npb = 4
        self.n_nodes = 4 * self.n_blocks

    def _grid(self):
        n2s, n1s = np.meshgrid(np.arange(self.n2_blocks), np.arange(self.n1_blocks), indexing="ij")
        return n1s.reshape(-1), n2s.reshape(-1)  # block id = n2 * n1_blocks + n1 (geometry.py:392-393)

    def centroid_node_vectors(self, angle):
        """geometry.py:382-395."""
        angle = _t(angle)
        n1s, n2s = self._grid()
        sign = _t((-1.0) ** (n1s + n2s))
        a = sign * angle
        v0 = (self.spacing - self.bond_length) / (2 * torch.cos(a))[:, None] * torch.stack([torch.cos(a), torch.sin(a)], -1)
        rots = rotation_matrix(torch.linspace(0.0, 3 * math.pi / 2, 4, dtype=F64))  # (4,2,2)
        return torch.einsum("kij,bj->bki", rots, v0)

    def block_centroids(self, angle=None):
        """geometry.py:397-404."""
        n1s, n2s = self._grid()
        return self.spacing * torch.stack([_t(n1s), _t(n2s)], -1)

    def bond_connectivity(self):
        """geometry.py:409-421."""
        n1b, n2b = self.n1_blocks, self.n2_blocks
        h = [[n1b * n2 * 4 + n1 * 4, n1b * n2 * 4 + (n1 + 1) * 4 + 2] for n2 in range(n2b) for n1 in range(n1b - 1)]
        v = [[n1b * n2 * 4 + n1 * 4 + 1, n1b * (n2 + 1) * 4 + n1 * 4 + 3] for n2 in range(n2b - 1) for n1 in range(n1b)]
        return np.array(h + v, dtype=np.int64).reshape(-1, 2)

    def reference_bond_vectors(self):
        """geometry.py:425-435."""
        n1b, n2b = self.n1_blocks, self.n2_blocks
        h = np.tile([self.bond_length, 0.0], ((n1b - 1) * n2b, 1))
        v = np.tile([0.0, self.bond_length], ((n2b - 1) * n1b, 1))
        return _t(np.concatenate([h, v]))


class QuadGeometry:
    """geometry.py:804-952."""

    def __init__(self, n1_blocks, n2_blocks, spacing=1.0, bond_length=0.1):
        self.n1_blocks, self.n2_blocks = n1_blocks, n2_blocks
        self.n1_cells, self.n2_cells = n1_blocks, n2_blocks
        self.spacing, self.bond_length = spacing, bond_length
        self.n_blocks = n1_blocks * n2_blocks
        self.n_npb = 4
        self.n_nodes = 4 * self.n_blocks

    def _grid(self):
        n2s, n1s = np.meshgrid(np.arange(self.n2_blocks), np.arange(self.n1_blocks), indexing="ij")
        return n1s.reshape(-1), n2s.reshape(-1)  # geometry.py:851-852

    def reference_node_vectors(self, horizontal_shift, vertical_shift):
        """geometry.py:832-854: nodes 0:+x 1:+y 2:-x 3:-y of the square grid plus the shifts."""
        hs, vs = _t(horizontal_shift), _t(vertical_shift)
        r = (self.spacing - self.bond_length) / 2
        v0s = torch.einsum("kij,j->ki", rotation_matrix(torch.linspace(0.0, 3 * math.pi / 2, 4, dtype=F64)),
                           torch.tensor([r, 0.0], dtype=F64))
        n1s, n2s = self._grid()
        sh = torch.stack([hs[n1s + 1, n2s], vs[n1s, n2s + 1], hs[n1s, n2s], vs[n1s, n2s]], 1)
        return v0s[None] + sh

    def centroid_node_vectors(self, horizontal_shift, vertical_shift):
        """geometry.py:856-868."""
        ref = self.reference_node_vectors(horizontal_shift, vertical_shift)
        return ref - polygon_centroid(ref)[:, None, :]

    def reference_points(self):
        n1s, n2s = self._grid()
        return self.spacing * torch.stack([_t(n1s), _t(n2s)], -1)

    def block_centroids(self, horizontal_shift, vertical_shift):
        """geometry.py:879-887."""
        ref = self.reference_node_vectors(horizontal_shift, vertical_shift)
        return self.reference_points() + polygon_centroid(ref)

    bond_connectivity = RotatedSquareGeometry.bond_connectivity  # geometry.py:892-904 (same pattern)
    reference_bond_vectors = RotatedSquareGeometry.reference_bond_vectors  # geometry.py:908-918

    def get_design_from_rotated_square(self, angle):
        """geometry.py:928-952."""
        r = self.spacing - self.bond_length

        def base(n1, n2):
            a = (-1.0) ** (n1 + n2) * angle
            return r / (2 * math.cos(a)) * np.array([math.cos(a), math.sin(a)]) - np.array([1.0, 0.0]) * r / 2

        hs = np.array([[base(n1, n2) for n2 in range(self.n2_blocks)] for n1 in range(self.n1_blocks + 1)])
        rot90 = np.array([[0.0, -1.0], [1.0, 0.0]])
        vs = np.array([[rot90 @ base(n1, n2) for n2 in range(self.n2_blocks + 1)] for n1 in range(self.n1_blocks)])
        return _t(hs), _t(vs)


class KagomeGeometry:
    """geometry.py:607-801 (non-periodic kagome; 2 triangles per cell)."""

    def __init__(self, n1_cells, n2_cells, direct_basis=None, bond_length=0.1):
        if direct_basis is None:
            direct_basis = np.array([[1.0, 0.0], [math.cos(math.pi / 3), math.sin(math.pi / 3)]])
        self.n1_cells, self.n2_cells = n1_cells, n2_cells
        self.direct_basis = _t(direct_basis)
        self.bond_length = bond_length
        self.n_cells = n1_cells * n2_cells
        self.n_bpc, self.n_npb = 2, 3
        self.n_blocks = 2 * self.n_cells
        self.n_nodes = 3 * self.n_blocks
        l = bond_length
        # geometry.py:642-644
        self.ref_int = _t([l * math.cos(math.pi / 6), l * math.sin(math.pi / 6)])
        self.ref_b1 = _t([0.0, -l])
        self.ref_b2 = _t([-l * math.cos(math.pi / 6), l * math.sin(math.pi / 6)])

    def reference_node_vectors(self, shifts_1, shifts_2, shifts_3):
        """geometry.py:646-686."""
        s1, s2, s3 = _t(shifts_1), _t(shifts_2), _t(shifts_3)
        a1, a2 = self.direct_basis[0], self.direct_basis[1]
        b1 = torch.stack([a1 / 2, a1 / 2 + a2 / 2, a2 / 2]) - 0.5 * torch.stack([self.ref_b1, self.ref_int, self.ref_b2])
        b2 = torch.stack([a1 / 2 + a2 / 2, a1 + a2 / 2, a1 / 2 + a2]) + 0.5 * torch.stack([self.ref_int, self.ref_b2, self.ref_b1])
        s11, s12 = s1[:-1], s1[1:]            # node (2), node (4)
        s21, s22 = s2[:, :-1], s2[:, 1:]      # node (0), node (5)
        blk1 = b1[None, None] + torch.stack([s21, s3, s11], 2)   # (n1, n2, 3, 2)
        blk2 = b2[None, None] + torch.stack([s3, s12, s22], 2)
        cells = torch.stack([blk1, blk2], 2)                     # (n1, n2, 2, 3, 2)
        return cells.permute(1, 0, 2, 3, 4).reshape(self.n_blocks, 3, 2)

    def centroid_node_vectors(self, shifts_1, shifts_2, shifts_3):
        """geometry.py:688-700."""
        ref = self.reference_node_vectors(shifts_1, shifts_2, shifts_3)
        return ref - polygon_centroid(ref)[:, None, :]

    def reference_points(self):
        """geometry.py:702-710."""
        n2s, n1s = np.meshgrid(np.arange(self.n2_cells), np.arange(self.n1_cells), indexing="ij")
        pts = _t(n1s.reshape(-1))[:, None] * self.direct_basis[0] + _t(n2s.reshape(-1))[:, None] * self.direct_basis[1]
        return torch.repeat_interleave(pts, 2, dim=0)

    def block_centroids(self, shifts_1, shifts_2, shifts_3):
        """geometry.py:712-724."""
        ref = self.reference_node_vectors(shifts_1, shifts_2, shifts_3)
        return self.reference_points() + polygon_centroid(ref)

    def bond_connectivity(self):
        """geometry.py:729-767."""
        n1c, n2c = self.n1_cells, self.n2_cells
        internal = [[1 + 6 * (n2 * n1c + n1), 3 + 6 * (n2 * n1c + n1)] for n2 in range(n2c) for n1 in range(n1c)]
        bnd1 = [[0 + 6 * ((n2 + 1) * n1c + n1), 5 + 6 * (n2 * n1c + n1)] for n2 in range(n2c - 1) for n1 in range(n1c)]
        bnd2 = [[2 + 6 * (n2 * n1c + n1 + 1), 4 + 6 * (n2 * n1c + n1)] for n2 in range(n2c) for n1 in range(n1c - 1)]
        return np.array(internal + bnd1 + bnd2, dtype=np.int64).reshape(-1, 2)

    def reference_bond_vectors(self):
        """geometry.py:771-789."""
        n1c, n2c = self.n1_cells, self.n2_cells
        return torch.cat([self.ref_int.expand(n1c * n2c, 2), self.ref_b1.expand(n1c * (n2c - 1), 2),
                          self.ref_b2.expand((n1c - 1) * n2c, 2)])
