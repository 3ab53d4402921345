"""Host-side geometry of the solver: design variables -> (block_centroids, centroid_node_vectors,
bond_connectivity, reference_bond_vectors), block inertia, DOF bookkeeping and the undeformed void
angles, each with the vector-Jacobian product the adjoint needs.  NumPy only.

Counterpart of the reference's ``difflexmm/geometry.py`` (run once per design, not per time step).
Derivatives of the small per-block polygon formulas are taken with the complex-step method
(exact to rounding, no subtraction), the index maps design -> nodes are linear and transpose to
scatter-adds.
"""
import math

import numpy as np

_CSTEP = 1e-40


def rotation_matrix(angle):
    """geometry.py:17-23."""
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s], [s, c]])


# -- polygons (geometry.py:71-160) -------------------------------------------------------------

def polygon_props(v):
    """area, centroid, polar moment about the centroid of CCW polygons v (..., n, 2).
    Works for complex input (used by the complex-step Jacobians): |S| is sign(Re S) * S."""
    x, y = v[..., 0], v[..., 1]
    x1, y1 = np.roll(x, 1, axis=-1), np.roll(y, 1, axis=-1)
    cr = x1 * y - y1 * x
    S = cr.sum(-1)
    area = np.where(S.real < 0, -1.0, 1.0) * S / 2                      # geometry.py:84
    cx = ((x1 + x) * cr).sum(-1) / (6 * area)                           # geometry.py:103-106
    cy = ((y1 + y) * cr).sum(-1) / (6 * area)
    ax, ay, bx, by = x1 - cx[..., None], y1 - cy[..., None], x - cx[..., None], y - cy[..., None]
    q = ax * ax + ax * bx + bx * bx + ay * ay + ay * by + by * by
    M = ((ax * by - ay * bx) * q).sum(-1) / 12
    ip = np.where(M.real < 0, -1.0, 1.0) * M                              # geometry.py:123-127
    return area, np.stack([cx, cy], -1), ip


# (area, polar moment) of centred node vectors produced by a lattice class in this process, keyed by the identity of the
# (read-only) array: compute_inertia on exactly that array then costs nothing (one polygon pass per design instead of three).
_PROPS_CACHE = {}


def _remember_props(cnv, area, ip):
    import weakref
    cnv.flags.writeable = False
    key = id(cnv)
    _PROPS_CACHE[key] = (weakref.ref(cnv, lambda _r, k=key: _PROPS_CACHE.pop(k, None)), area, ip)


def _recall_props(v):
    hit = _PROPS_CACHE.get(id(v))
    if hit is not None and hit[0]() is v:
        return hit[1], hit[2]
    return None


def current_coordinates(vertices, centroids, angles, displacements):
    """geometry.py:26-36: deformed vertex coordinates, (n_blocks, n_npb, 2): every block's vertices rotated by its angle, then moved
    to centroid + displacement."""
    v, ang = np.asarray(vertices, dtype=float), np.asarray(angles, dtype=float)
    c, s = np.cos(ang), np.sin(ang)
    Q = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2)
    return np.einsum("bij,bnj->bni", Q, v) + (np.asarray(centroids, dtype=float) + np.asarray(displacements, dtype=float))[:, None, :]


def get_point_ids_in_bounding_box(points, bounding_box):
    """geometry.py:39-53: indices of the points inside [[x_min, y_min], [x_max, y_max]] (borders included)."""
    p, bb = np.asarray(points), np.asarray(bounding_box)
    return np.where((p[:, 0] >= bb[0, 0]) & (p[:, 0] <= bb[1, 0]) & (p[:, 1] >= bb[0, 1]) & (p[:, 1] <= bb[1, 1]))[0]


def get_point_ids_in_circle(points, center, radius):
    """geometry.py:56-68."""
    return np.where(np.linalg.norm(np.asarray(points) - np.asarray(center), axis=1) <= radius)[0]


def compute_xy_limits(points):
    """geometry.py:256-267: [[x_min, x_max], [y_min, y_max]]."""
    p = np.asarray(points)
    return np.array([p.min(axis=0), p.max(axis=0)]).T


def polygon_area(v):
    return polygon_props(np.asarray(v, dtype=float))[0]


def polygon_centroid(v):
    return polygon_props(np.asarray(v, dtype=float))[1]


def polygon_polar_moment(v):
    return polygon_props(np.asarray(v, dtype=float))[2]


def polygons_geometric_properties(vertices):
    """geometry.py:131-142: (centroid, area, polar moment of area) of an array of polygons."""
    a, c, j = polygon_props(np.asarray(vertices, dtype=float))
    return c, a, j


def polygon_props_jac(v):
    """Jacobians of (area, centroid, polar moment) w.r.t. the vertices, per polygon.
    Returns dA (nb,n,2), dC (nb,2,n,2), dI (nb,n,2)."""
    v = np.asarray(v, dtype=float)
    nb, n, _ = v.shape
    dA = np.empty((nb, n, 2)); dC = np.empty((nb, 2, n, 2)); dI = np.empty((nb, n, 2))
    for k in range(n):
        for c in range(2):
            vc = v.astype(complex)
            vc[:, k, c] += 1j * _CSTEP
            a, cen, ip = polygon_props(vc)
            dA[:, k, c] = a.imag / _CSTEP
            dC[:, :, k, c] = cen.imag / _CSTEP
            dI[:, k, c] = ip.imag / _CSTEP
    return dA, dC, dI


def polygon_props_vjp(v, area_bar=None, cen_bar=None, ip_bar=None):
    """Cotangent of the vertices v (..., n, 2) for cotangents of polygon_props' (area, centroid, polar moment): the closed-form
    derivatives (the complex-step Jacobians of polygon_props_jac, 2n polygon passes per design, were most of the per-design host
    time of the design loop; they remain as the checker in the tests).
    With x1 = x_{k-1}, cr_k = x1 y - y1 x, S = sum cr, A = |S|/2:  dA/dx_k = sgn (y_{k+1} - y_{k-1}) / 2, dA/dy_k = sgn (x_{k-1} - x_{k+1}) / 2;
    c = N / (3 sgn S) with N_x = sum (x1 + x) cr, N_y = sum (y1 + y) cr;  the polar moment about the centroid is stationary
    w.r.t. the reference point at the centroid, so dI = dM at fixed c, M = sum (a x b)(|a|^2 + a.b + |b|^2) / 12, a = p_{k-1} - c, b = p_k - c."""
    v = np.asarray(v, dtype=float)
    x, y = v[..., 0], v[..., 1]
    x1, y1 = np.roll(x, 1, axis=-1), np.roll(y, 1, axis=-1)          # previous vertex
    xn, yn = np.roll(x, -1, axis=-1), np.roll(y, -1, axis=-1)        # next vertex
    cr = x1 * y - y1 * x
    crn = np.roll(cr, -1, axis=-1)                                   # cr_{k+1} = x y_{k+1} - y x_{k+1}
    S = cr.sum(-1)
    sgn = np.where(S < 0, -1.0, 1.0)
    dAx, dAy = 0.5 * sgn[..., None] * (yn - y1), 0.5 * sgn[..., None] * (x1 - xn)
    out = np.zeros_like(v)
    if area_bar is not None:
        ab = np.asarray(area_bar, dtype=float)[..., None]
        out[..., 0] += ab * dAx
        out[..., 1] += ab * dAy
    if cen_bar is not None or ip_bar is not None:
        area = 0.5 * sgn * S
        cx = ((x1 + x) * cr).sum(-1) / (6 * area)
        cy = ((y1 + y) * cr).sum(-1) / (6 * area)
    gx = gy = None
    if cen_bar is not None:
        gb = np.asarray(cen_bar, dtype=float)
        gx, gy = gb[..., 0][..., None], gb[..., 1][..., None]
    if ip_bar is not None:
        ax, ay, bx, by = x1 - cx[..., None], y1 - cy[..., None], x - cx[..., None], y - cy[..., None]
        q = ax * ax + ax * bx + bx * bx + ay * ay + ay * by + by * by
        w = ax * by - ay * bx
        M = (w * q).sum(-1) / 12
        sm = (np.where(M < 0, -1.0, 1.0) * np.asarray(ip_bar, dtype=float) / 12)[..., None]
        # vertex k is b of term k and a of term k + 1
        tbx, tby = -ay * q + w * (ax + 2 * bx), ax * q + w * (ay + 2 * by)
        tax, tay = by * q + w * (2 * ax + bx), -bx * q + w * (2 * ay + by)
        out[..., 0] += sm * (tbx + np.roll(tax, -1, axis=-1))
        out[..., 1] += sm * (tby + np.roll(tay, -1, axis=-1))
        # through the reference point: zero for counter-clockwise polygons (c is then the centroid, where the polar moment is
        # stationary); for clockwise input polygon_props divides by |S| and returns the mirrored point, like the reference
        ex, ey = -(sm * (tax + tbx)).sum(-1, keepdims=True), -(sm * (tay + tby)).sum(-1, keepdims=True)
        gx, gy = (ex if gx is None else gx + ex), (ey if gy is None else gy + ey)
    if gx is not None:
        dNx_dx = cr - (x1 + x) * y1 + crn + (x + xn) * yn
        dNx_dy = (x1 + x) * x1 - (x + xn) * xn
        dNy_dx = -(y1 + y) * y1 + (y + yn) * yn
        dNy_dy = cr + (y1 + y) * x1 + crn - (y + yn) * xn
        i6a = (1.0 / (6 * area))[..., None]
        ia = (1.0 / area)[..., None]
        out[..., 0] += gx * (dNx_dx * i6a - cx[..., None] * dAx * ia) + gy * (dNy_dx * i6a - cy[..., None] * dAx * ia)
        out[..., 1] += gx * (dNx_dy * i6a - cx[..., None] * dAy * ia) + gy * (dNy_dy * i6a - cy[..., None] * dAy * ia)
    return out


def compute_inertia(vertices, density):
    """geometry.py:144-160 -> (n_blocks, 3) = [rho A, rho A, rho I_p]."""
    hit = _recall_props(vertices)
    if hit is None:
        area, _, ip = polygon_props(np.asarray(vertices, dtype=float))
    else:
        area, ip = hit
    m = density * area
    return np.column_stack((m, m, density * ip))


def compute_inertia_vjp(vertices, density, inertia_bar):
    """(vertices_bar, density_bar) for a cotangent of compute_inertia's result."""
    vertices = np.asarray(vertices, dtype=float)
    hit = _recall_props(vertices)
    area, ip = hit if hit is not None else polygon_props(vertices)[::2]
    gm = inertia_bar[:, 0] + inertia_bar[:, 1]
    rho = np.broadcast_to(np.asarray(density, dtype=float), area.shape)
    v_bar = polygon_props_vjp(vertices, area_bar=rho * gm, ip_bar=rho * inertia_bar[:, 2])
    rho_bar = gm * area + inertia_bar[:, 2] * ip
    return v_bar, (rho_bar.sum() if np.ndim(density) == 0 else rho_bar)


def DOFsInfo(n_blocks, constrained_block_DOF_pairs):
    """geometry.py:163-178, vectorised (O(n) instead of the reference's O(n_dof * n_constrained))."""
    pairs = np.asarray(constrained_block_DOF_pairs, dtype=np.int64).reshape(-1, 2)
    constrained = pairs[:, 0] * 3 + pairs[:, 1]
    all_ids = np.arange(n_blocks * 3)
    mask = np.ones(n_blocks * 3, dtype=bool)
    mask[constrained] = False
    return all_ids[mask], constrained, all_ids


def compute_edge_lengths(cnv):
    """geometry.py:205-218."""
    return np.linalg.norm(np.roll(cnv, 1, axis=1) - cnv, axis=2)


# -- void angles of the undeformed design (energy.py:204-219 + geometry.py:181-253 at u = 0) -----

_EDGE_INDEX_CACHE = {}


def _edge_index(bonds, n):
    """(b1, l1, next, prev, b2, l2, next, prev) of every bond; cached per connectivity array (it is static per lattice and
    the integer divisions below were a third of the per-design host time)."""
    key = (id(bonds), np.shape(bonds), n)
    hit = _EDGE_INDEX_CACHE.get(key)
    if hit is not None and hit[0] is bonds:
        return hit[1]
    bb = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    b1, l1 = bb[:, 0] // n, bb[:, 0] % n
    b2, l2 = bb[:, 1] // n, bb[:, 1] % n
    idx = (b1, l1, (l1 + 1) % n, (l1 - 1) % n, b2, l2, (l2 + 1) % n, (l2 - 1) % n)
    if len(_EDGE_INDEX_CACHE) > 64:
        _EDGE_INDEX_CACHE.clear()
    _EDGE_INDEX_CACHE[key] = (bonds, idx)       # the array is kept alive, so its id cannot be reused while cached
    return idx


def _edge_pairs(cnv, bonds):
    n = cnv.shape[1]
    idx = _edge_index(bonds, n)
    b1, l1, b2, l2 = idx[0], idx[1], idx[4], idx[5]
    e1p = cnv[b1, idx[2]] - cnv[b1, l1]   # edge node1 -> next node   (block_1_node_1, unnormalised)
    e1m = cnv[b1, idx[3]] - cnv[b1, l1]   # edge node1 -> previous node
    e2p = cnv[b2, idx[6]] - cnv[b2, l2]
    e2m = cnv[b2, idx[7]] - cnv[b2, l2]
    return idx, e1p, e1m, e2p, e2m


def compute_edge_unit_vectors(current_block_nodes, node_id):
    """geometry.py:181-202: unit vectors from node ``node_id`` (global index, or an array of them) to the next and to the previous node
    of its block."""
    nodes = np.asarray(current_block_nodes, dtype=float)
    n = nodes.shape[1]
    nid = np.asarray(node_id, dtype=np.int64)
    b, l = nid // n, nid % n
    u1 = nodes[b, (l + 1) % n] - nodes[b, l]
    u2 = nodes[b, (l - 1) % n] - nodes[b, l]
    return u1 / np.linalg.norm(u1, axis=-1, keepdims=True), u2 / np.linalg.norm(u2, axis=-1, keepdims=True)


def angle_between_unit_vectors(u1, u2):
    """geometry.py:221-231: signed angle from u1 to u2 (counter-clockwise positive), in [-pi, pi]."""
    u1, u2 = np.asarray(u1, dtype=float), np.asarray(u2, dtype=float)
    return np.arctan2(u1[..., 0] * u2[..., 1] - u1[..., 1] * u2[..., 0], u1[..., 0] * u2[..., 0] + u1[..., 1] * u2[..., 1])


def _angle(u, w):
    return np.arctan2(u[:, 0] * w[:, 1] - u[:, 1] * w[:, 0], u[:, 0] * w[:, 0] + u[:, 1] * w[:, 1])


def void_angles0(cnv, bonds):
    """(n_bonds, 2): the two void angles of every bond in the undeformed design.  During the motion the
    angles are  wrap(phi1 + th_A - th_B), wrap(phi2 + th_B - th_A)  (rigid blocks: translations cancel)."""
    cnv = np.asarray(cnv, dtype=float)
    _, e1p, e1m, e2p, e2m = _edge_pairs(cnv, bonds)
    return np.stack([_angle(e2m, e1p), _angle(e1m, e2p)], 1)   # geometry.py:248-249


def void_angles0_vjp(cnv, bonds, phi_bar):
    """cnv_bar for a cotangent (n_bonds, 2) of void_angles0."""
    cnv = np.asarray(cnv, dtype=float)
    (b1, l1, n1, p1, b2, l2, n2, p2), e1p, e1m, e2p, e2m = _edge_pairs(cnv, bonds)
    out = np.zeros_like(cnv)

    def back(u, w, gbar):
        # phi = atan2(u x w, u . w): dphi/du = -perp(u)/|u|^2 , dphi/dw = perp(w)/|w|^2, perp(a) = (-a_y, a_x)
        du = -np.stack([-u[:, 1], u[:, 0]], 1) / (u ** 2).sum(1)[:, None] * gbar[:, None]
        dw = np.stack([-w[:, 1], w[:, 0]], 1) / (w ** 2).sum(1)[:, None] * gbar[:, None]
        return du, dw

    du, dw = back(e2m, e1p, phi_bar[:, 0])      # phi1: u = e2m, w = e1p
    np.add.at(out, (b2, p2), du); np.add.at(out, (b2, l2), -du)
    np.add.at(out, (b1, n1), dw); np.add.at(out, (b1, l1), -dw)
    du, dw = back(e1m, e2p, phi_bar[:, 1])      # phi2: u = e1m, w = e2p
    np.add.at(out, (b1, p1), du); np.add.at(out, (b1, l1), -du)
    np.add.at(out, (b2, n2), dw); np.add.at(out, (b2, l2), -dw)
    return out


def compute_edge_angles(nodes, bonds):
    """geometry.py:234-253 on explicit node positions (n_blocks, n_npb, 2): void_1, void_2, block_1, block_2."""
    nodes = np.asarray(nodes, dtype=float)
    _, e1p, e1m, e2p, e2m = _edge_pairs(nodes, bonds)
    return _angle(e2m, e1p), _angle(e1m, e2p), _angle(e1p, e1m), _angle(e2p, e2m)


# -- the same maps, fused, in native code (include/dfx.h: dfx_design_forward / dfx_design_vjp) -------------------------------------

class NativeDesignMap:
    """Design -> (block_centroids, centroid_node_vectors, inertia, undeformed void angles) and the cotangent of all of it in ONE native loop
    over the blocks (``csrc/dfx_design.h``, host code in libdfx -- no device involved), for any lattice class whose nodes are "a static base
    vector + one row of the design" (QuadGeometry, KagomeGeometry: geometry.py:607-952).  The static tables are derived from the class itself:
    ``base`` = the node vectors of the zero design, ``gather`` = which design row each node takes, found by pushing the row numbers through
    ``reference_node_vectors``.  The NumPy functions of this module stay as the checker (tests/test_host_helpers.py, 1e-13)."""

    def __init__(self, geometry, bonds=None):
        shapes = geometry.design_shapes()
        self.shapes = shapes
        self.sizes = [int(np.prod(sh[:-1])) for sh in shapes]
        self.n_design = int(sum(self.sizes))
        zeros = [np.zeros(sh) for sh in shapes]
        base = np.ascontiguousarray(geometry.reference_node_vectors(*zeros), dtype=np.float64)
        rows, start = [], 0
        for sh, n in zip(shapes, self.sizes):
            a = np.zeros(sh)
            a.reshape(-1, 2)[:, 0] = np.arange(start, start + n)
            rows.append(a)
            start += n
        gather = np.rint(geometry.reference_node_vectors(*rows)[..., 0] - base[..., 0]).astype(np.int32)
        self.base, self.gather = base, np.ascontiguousarray(gather)
        self.ref_points = np.ascontiguousarray(geometry.reference_points(), dtype=np.float64)
        self.n_blocks, self.n_npb = base.shape[0], base.shape[1]
        self.bonds = None if bonds is None else np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 2)

    def _map(self):
        from ._binding import _dp, _ip, dfx_design_map
        m = dfx_design_map()
        m.n_blocks, m.n_npb, m.n_design = self.n_blocks, self.n_npb, self.n_design
        m.n_bonds = 0 if self.bonds is None else len(self.bonds)
        m.base, m.gather, m.ref_points = self.base.ctypes.data_as(_dp), self.gather.ctypes.data_as(_ip), self.ref_points.ctypes.data_as(_dp)
        m.bonds = None if self.bonds is None else self.bonds.ctypes.data_as(_ip)
        return m

    def flatten(self, designs):
        """list of design tuples -> (batch, n_design, 2)."""
        return np.ascontiguousarray(np.stack([np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in d]) for d in designs]))

    def unflatten(self, flat):
        out, start = [], 0
        for sh, n in zip(self.shapes, self.sizes):
            out.append(flat[start:start + n].reshape(sh))
            start += n
        return tuple(out)

    def forward(self, lib, designs, density, void_angles=True):
        """-> block_centroids (B, n, 2), cnv (B, n, npb, 2), inertia (B, n, 3), void_angle0 (B, n_bonds, 2) or None."""
        import ctypes as C
        from ._binding import _ptr
        flat = self.flatten(designs)
        B, nb, n = len(designs), self.n_blocks, self.n_npb
        cen, cnv, inertia = np.empty((B, nb, 2)), np.empty((B, nb, n, 2)), np.empty((B, nb, 3))
        va = np.empty((B, len(self.bonds), 2)) if (void_angles and self.bonds is not None) else None
        m = self._map()
        if lib.dfx_design_forward(C.byref(m), _ptr(flat), B, float(density), _ptr(cen), _ptr(cnv), _ptr(inertia), _ptr(va)) != 0:
            raise RuntimeError("dfx_design_forward failed")
        return cen, cnv, inertia, va

    def vjp(self, lib, designs, density, cnv_bar, centroid_bar=None, inertia_bar=None, void_bar=None):
        """Cotangents (batch-leading arrays; None = zero) -> list of design-gradient tuples."""
        import ctypes as C
        from ._binding import _ptr, _f64
        flat = self.flatten(designs)
        B = len(designs)
        out = np.empty_like(flat)
        m = self._map()
        args = [None if a is None else _f64(a) for a in (cnv_bar, centroid_bar, inertia_bar, void_bar)]
        if lib.dfx_design_vjp(C.byref(m), _ptr(flat), B, float(density), _ptr(args[0]), _ptr(args[1]), _ptr(args[2]), _ptr(args[3]), _ptr(out)) != 0:
            raise RuntimeError("dfx_design_vjp failed")
        return [self.unflatten(out[i]) for i in range(B)]


# -- lattices ---------------------------------------------------------------------------------------

class Geometry:
    """Common part of the lattice classes (reference: ``Geometry``/``LatticeGeometry``, geometry.py:272-351)."""
    n_blocks: int
    n_npb: int

    def get_parametrization(self):
        """geometry.py:319-328."""
        return self.block_centroids, self.centroid_node_vectors, self.bond_connectivity, self.reference_bond_vectors

    def get_reference_geometry(self, *design):
        return self.centroid_node_vectors(*design) + self.block_centroids(*design)[:, None, :]

    # centred node vectors / centroids from the un-centred reference node vectors
    def _centre(self, ref):
        area, cen, ip = polygon_props(ref)
        cnv = ref - cen[:, None, :]
        _remember_props(cnv, area, ip)      # the polar moment is about the centroid: translation-invariant
        return cnv, cen

    def geometry_from_design(self, *design):
        """(block_centroids, centroid_node_vectors) from one polygon pass (the two reference callables each do their own)."""
        cnv, cen = self._centre(self.reference_node_vectors(*design))
        return self.reference_points() + cen, cnv

    def _centre_vjp(self, ref, cnv_bar, centroid_bar, props_bar=None):
        """props_bar = (area_bar, polar_moment_bar) of the centred polygons, folded into the same polygon pass: area and polar moment about the
        centroid do not change under translation, so their vertex derivatives on ``ref`` are those on ``ref - centroid`` and sum to zero over
        a polygon (no contribution to the centroid's cotangent) -- one pass per design instead of two (compute_inertia_vjp + this)."""
        g = (centroid_bar if centroid_bar is not None else 0.0) - cnv_bar.sum(1)   # cotangent of the centroid
        if props_bar is None:
            return cnv_bar + polygon_props_vjp(ref, cen_bar=g)
        return cnv_bar + polygon_props_vjp(ref, area_bar=props_bar[0], cen_bar=g, ip_bar=props_bar[1])


def _square_grid(n1, n2):
    n2s, n1s = np.meshgrid(np.arange(n2), np.arange(n1), indexing="ij")
    return n1s.reshape(-1), n2s.reshape(-1)


def _quad_bonds(n1b, n2b):
    h = [[n1b * n2 * 4 + n1 * 4, n1b * n2 * 4 + (n1 + 1) * 4 + 2] for n2 in range(n2b) for n1 in range(n1b - 1)]
    v = [[n1b * n2 * 4 + n1 * 4 + 1, n1b * (n2 + 1) * 4 + n1 * 4 + 3] for n2 in range(n2b - 1) for n1 in range(n1b)]
    return np.array(h + v, dtype=np.int32).reshape(-1, 2)


def _quad_ref_vectors(n1b, n2b, l):
    return np.concatenate([np.tile([l, 0.0], ((n1b - 1) * n2b, 1)), np.tile([0.0, l], ((n2b - 1) * n1b, 1))])


class QuadGeometry(Geometry):
    """Aperiodic lattice of quadrilaterals, design = (horizontal_shifts (n1+1,n2,2), vertical_shifts (n1,n2+1,2)).
    Reference: geometry.py:804-952 (block id = n2*n1_blocks + n1; nodes 0:+x 1:+y 2:-x 3:-y)."""

    def __init__(self, n1_blocks, n2_blocks, spacing=1.0, bond_length=0.1):
        self.n1_blocks, self.n2_blocks = int(n1_blocks), int(n2_blocks)
        self.n1_cells, self.n2_cells = self.n1_blocks, self.n2_blocks
        self.spacing, self.bond_length = float(spacing), float(bond_length)
        self.n_blocks = self.n1_blocks * self.n2_blocks
        self.n_npb = 4
        self.n_nodes = 4 * self.n_blocks
        self._n1s, self._n2s = _square_grid(self.n1_blocks, self.n2_blocks)
        r = (self.spacing - self.bond_length) / 2
        self._v0s = np.array([[r, 0.0], [0.0, r], [-r, 0.0], [0.0, -r]])

    def design_shapes(self):
        return (self.n1_blocks + 1, self.n2_blocks, 2), (self.n1_blocks, self.n2_blocks + 1, 2)

    def reference_node_vectors(self, horizontal_shifts, vertical_shifts):
        hs, vs = np.asarray(horizontal_shifts, dtype=float), np.asarray(vertical_shifts, dtype=float)
        n1, n2 = self._n1s, self._n2s
        return self._v0s[None] + np.stack([hs[n1 + 1, n2], vs[n1, n2 + 1], hs[n1, n2], vs[n1, n2]], 1)

    def centroid_node_vectors(self, horizontal_shifts, vertical_shifts):
        return self._centre(self.reference_node_vectors(horizontal_shifts, vertical_shifts))[0]

    def reference_points(self):
        return self.spacing * np.stack([self._n1s, self._n2s], -1).astype(float)

    def block_centroids(self, horizontal_shifts, vertical_shifts):
        return self.reference_points() + self._centre(self.reference_node_vectors(horizontal_shifts, vertical_shifts))[1]

    def bond_connectivity(self):
        return _quad_bonds(self.n1_blocks, self.n2_blocks)

    def reference_bond_vectors(self):
        return _quad_ref_vectors(self.n1_blocks, self.n2_blocks, self.bond_length)

    def vjp(self, design, cnv_bar, centroid_bar=None, props_bar=None):
        """Cotangents of (centroid_node_vectors, block_centroids[, (area, polar moment)]) -> cotangents of the design tuple."""
        hs, vs = design
        ref_bar = self._centre_vjp(self.reference_node_vectors(hs, vs), cnv_bar, centroid_bar, props_bar)
        n1, n2 = self._n1s, self._n2s
        hb = np.zeros(self.design_shapes()[0]); vb = np.zeros(self.design_shapes()[1])
        np.add.at(hb, (n1 + 1, n2), ref_bar[:, 0]); np.add.at(vb, (n1, n2 + 1), ref_bar[:, 1])
        np.add.at(hb, (n1, n2), ref_bar[:, 2]); np.add.at(vb, (n1, n2), ref_bar[:, 3])
        return hb, vb

    def get_design_from_rotated_square(self, angle):
        """geometry.py:928-952."""
        r = self.spacing - self.bond_length

        def base(n1, n2):
            a = (-1.0) ** (n1 + n2) * angle
            return r / (2 * math.cos(a)) * np.array([math.cos(a), math.sin(a)]) - np.array([r / 2, 0.0])

        hs = np.array([[base(n1, n2) for n2 in range(self.n2_blocks)] for n1 in range(self.n1_blocks + 1)])
        rot = rotation_matrix(np.pi / 2)
        vs = np.array([[rot @ base(n1, n2) for n2 in range(self.n2_blocks + 1)] for n1 in range(self.n1_blocks)])
        return hs, vs


class RotatedSquareGeometry(Geometry):
    """Rotated-squares lattice, design = (angle,).  Reference: geometry.py:354-443."""

    def __init__(self, n1_cells, n2_cells, spacing=1.0, bond_length=0.1):
        self.n1_cells, self.n2_cells = int(n1_cells), int(n2_cells)
        self.spacing, self.bond_length = float(spacing), float(bond_length)
        self.n1_blocks, self.n2_blocks = 2 * self.n1_cells, 2 * self.n2_cells
        self.n_blocks = self.n1_blocks * self.n2_blocks
        self.n_npb = 4
        self.n_nodes = 4 * self.n_blocks
        self._n1s, self._n2s = _square_grid(self.n1_blocks, self.n2_blocks)

    def centroid_node_vectors(self, angle):
        a = (-1.0) ** (self._n1s + self._n2s) * angle
        v0 = ((self.spacing - self.bond_length) / (2 * np.cos(a)))[:, None] * np.stack([np.cos(a), np.sin(a)], -1)
        rots = np.stack([rotation_matrix(t) for t in np.linspace(0.0, 3 * np.pi / 2, 4)])
        return np.einsum("kij,bj->bki", rots, v0)

    def block_centroids(self, angle=None):
        return self.spacing * np.stack([self._n1s, self._n2s], -1).astype(float)

    def bond_connectivity(self):
        return _quad_bonds(self.n1_blocks, self.n2_blocks)

    def reference_bond_vectors(self):
        return _quad_ref_vectors(self.n1_blocks, self.n2_blocks, self.bond_length)


class KagomeGeometry(Geometry):
    """Non-periodic kagome lattice (2 triangles per cell), design = (shifts_1 (n1+1,n2,2), shifts_2 (n1,n2+1,2),
    shifts_3 (n1,n2,2)).  Reference: geometry.py:607-801."""

    def __init__(self, n1_cells, n2_cells, direct_basis=None, bond_length=0.1):
        if direct_basis is None:
            direct_basis = np.array([[1.0, 0.0], [math.cos(math.pi / 3), math.sin(math.pi / 3)]])
        self.n1_cells, self.n2_cells = int(n1_cells), int(n2_cells)
        self.direct_basis = np.asarray(direct_basis, dtype=float)
        self.bond_length = float(bond_length)
        self.n_cells = self.n1_cells * self.n2_cells
        self.n_bpc, self.n_npb = 2, 3
        self.n_blocks = 2 * self.n_cells
        self.n_nodes = 3 * self.n_blocks
        l = self.bond_length
        self._ri = l * np.array([math.cos(math.pi / 6), math.sin(math.pi / 6)])
        self._r1 = l * np.array([0.0, -1.0])
        self._r2 = l * np.array([-math.cos(math.pi / 6), math.sin(math.pi / 6)])
        a1, a2 = self.direct_basis
        self._b1 = np.array([a1 / 2, a1 / 2 + a2 / 2, a2 / 2]) - 0.5 * np.array([self._r1, self._ri, self._r2])
        self._b2 = np.array([a1 / 2 + a2 / 2, a1 + a2 / 2, a1 / 2 + a2]) + 0.5 * np.array([self._ri, self._r2, self._r1])

    def design_shapes(self):
        n1, n2 = self.n1_cells, self.n2_cells
        return (n1 + 1, n2, 2), (n1, n2 + 1, 2), (n1, n2, 2)

    def reference_node_vectors(self, shifts_1, shifts_2, shifts_3):
        s1, s2, s3 = (np.asarray(s, dtype=float) for s in (shifts_1, shifts_2, shifts_3))
        blk1 = self._b1[None, None] + np.stack([s2[:, :-1], s3, s1[:-1]], 2)     # nodes (0) (1) (2)
        blk2 = self._b2[None, None] + np.stack([s3, s1[1:], s2[:, 1:]], 2)       # nodes (3) (4) (5)
        cells = np.stack([blk1, blk2], 2)                                          # (n1, n2, 2, 3, 2)
        return cells.transpose(1, 0, 2, 3, 4).reshape(self.n_blocks, 3, 2)

    def centroid_node_vectors(self, shifts_1, shifts_2, shifts_3):
        return self._centre(self.reference_node_vectors(shifts_1, shifts_2, shifts_3))[0]

    def reference_points(self):
        n1s, n2s = _square_grid(self.n1_cells, self.n2_cells)
        pts = n1s[:, None] * self.direct_basis[0] + n2s[:, None] * self.direct_basis[1]
        return np.repeat(pts, 2, axis=0)

    def block_centroids(self, shifts_1, shifts_2, shifts_3):
        return self.reference_points() + self._centre(self.reference_node_vectors(shifts_1, shifts_2, shifts_3))[1]

    def bond_connectivity(self):
        n1c, n2c = self.n1_cells, self.n2_cells
        internal = [[1 + 6 * (n2 * n1c + n1), 3 + 6 * (n2 * n1c + n1)] for n2 in range(n2c) for n1 in range(n1c)]
        bnd1 = [[6 * ((n2 + 1) * n1c + n1), 5 + 6 * (n2 * n1c + n1)] for n2 in range(n2c - 1) for n1 in range(n1c)]
        bnd2 = [[2 + 6 * (n2 * n1c + n1 + 1), 4 + 6 * (n2 * n1c + n1)] for n2 in range(n2c) for n1 in range(n1c - 1)]
        return np.array(internal + bnd1 + bnd2, dtype=np.int32).reshape(-1, 2)

    def reference_bond_vectors(self):
        n1c, n2c = self.n1_cells, self.n2_cells
        return np.concatenate([np.tile(self._ri, (n1c * n2c, 1)), np.tile(self._r1, (n1c * (n2c - 1), 1)),
                               np.tile(self._r2, ((n1c - 1) * n2c, 1))])

    def vjp(self, design, cnv_bar, centroid_bar=None, props_bar=None):
        ref_bar = self._centre_vjp(self.reference_node_vectors(*design), cnv_bar, centroid_bar, props_bar)
        n1, n2 = self.n1_cells, self.n2_cells
        cells = ref_bar.reshape(n2, n1, 2, 3, 2).transpose(1, 0, 2, 3, 4)        # (n1, n2, 2, 3, 2)
        s1b, s2b, s3b = (np.zeros(s) for s in self.design_shapes())
        s2b[:, :-1] += cells[:, :, 0, 0]; s3b += cells[:, :, 0, 1]; s1b[:-1] += cells[:, :, 0, 2]
        s3b += cells[:, :, 1, 0]; s1b[1:] += cells[:, :, 1, 1]; s2b[:, 1:] += cells[:, :, 1, 2]
        return s1b, s2b, s3b
