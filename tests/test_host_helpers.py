"""Small host-side names of the reference's geometry / kinematics / utils modules that notebooks and scripts import
(geometry.py:26-68, 131-142, 181-231, 256-267; kinematics.py:13-81; utils.py:46) against the oracle's restatements or closed forms."""
import math

import numpy as np
import pytest
import torch

from difflexmm_amd import geometry as G
from difflexmm_amd import kinematics as K
from difflexmm_amd import utils as U
from oracle import ref_dynamics as OD
from oracle import ref_energy as OE
from oracle import ref_geometry as OG


def test_geometry_helpers():
    rng = np.random.default_rng(0)
    g = G.QuadGeometry(4, 3, 15.0, 2.25)
    design = tuple(rng.uniform(-0.5, 0.5, s) for s in g.design_shapes())
    cnv, cen = g.centroid_node_vectors(*design), g.block_centroids(*design)
    ang, disp = rng.uniform(-0.4, 0.4, 12), rng.normal(size=(12, 2))
    cur = G.current_coordinates(cnv, cen, ang, disp)
    b, k = 7, 2
    assert np.allclose(cur[b, k], G.rotation_matrix(ang[b]) @ cnv[b, k] + cen[b] + disp[b], atol=1e-14)
    nodes = np.array([5, 17, 30, 47])
    u1, u2 = G.compute_edge_unit_vectors(cur, nodes)
    o1, o2 = OG.compute_edge_unit_vectors(torch.tensor(cur), nodes)
    assert np.allclose(u1, o1.numpy(), atol=1e-14) and np.allclose(u2, o2.numpy(), atol=1e-14)
    assert np.allclose(G.angle_between_unit_vectors(u1, u2), OG.angle_between_unit_vectors(o1, o2).numpy(), atol=1e-14)
    single = G.compute_edge_unit_vectors(cur, 17)
    assert single[0].shape == (2,) and np.allclose(single[0], u1[1])
    assert abs(G.angle_between_unit_vectors(np.array([1.0, 0.0]), np.array([0.0, 1.0])) - math.pi / 2) < 1e-15
    pts = cur.reshape(-1, 2)
    lim = G.compute_xy_limits(pts)
    assert lim.shape == (2, 2) and lim[0, 0] == pts[:, 0].min() and lim[1, 1] == pts[:, 1].max()
    ids = G.get_point_ids_in_bounding_box(cen, np.array([[10.0, 10.0], [40.0, 25.0]]))
    assert set(ids) == {i for i, c in enumerate(cen) if 10 <= c[0] <= 40 and 10 <= c[1] <= 25} and len(ids) > 0
    ids = G.get_point_ids_in_circle(cen, cen[5], 15.5)
    assert 5 in ids and set(ids) == {i for i, c in enumerate(cen) if np.hypot(*(c - cen[5])) <= 15.5}
    c, a, j = G.polygons_geometric_properties(cnv)
    assert np.allclose(a, G.polygon_area(cnv)) and np.allclose(c, G.polygon_centroid(cnv)) and np.allclose(j, G.polygon_polar_moment(cnv))


def test_kinematics_module_and_solution_type():
    rng = np.random.default_rng(1)
    g = G.QuadGeometry(3, 3, 15.0, 2.25)
    og = OG.QuadGeometry(3, 3, 15.0, 2.25)
    pairs = np.array([[0, 0], [0, 1], [4, 2]])
    free, con, _ = G.DOFsInfo(9, pairs)
    fn = lambda t, amplitude: amplitude * t * np.array([1.0, 0.0, -2.0])          # noqa: E731
    kin = K.build_constrained_kinematics(g, pairs, fn)
    okin = OD.build_constrained_kinematics(og, pairs, lambda t, amplitude: amplitude * t * torch.tensor([1.0, 0.0, -2.0], dtype=torch.float64))
    q = rng.normal(size=len(free))
    a = kin(q, 0.3, dict(amplitude=2.0))
    b = okin(torch.tensor(q), 0.3, dict(amplitude=torch.tensor(2.0, dtype=torch.float64))).numpy()
    assert a.shape == (9, 3) and np.allclose(a, b, atol=1e-15) and a[0, 0] == 0.6 and a[4, 2] == -1.2
    assert np.array_equal(K.build_constrained_kinematics(g, np.zeros((0, 2), dtype=int))(np.arange(27.0), 0.0).ravel(), np.arange(27.0))
    cnv = g.centroid_node_vectors(*[rng.uniform(-0.3, 0.3, s) for s in g.design_shapes()])
    u = rng.normal(size=(9, 3)) * np.array([1.0, 1.0, 0.3])
    assert np.allclose(K.block_to_node_kinematics(u, cnv), OE.block_to_node_kinematics(torch.tensor(u), torch.tensor(cnv)).numpy(), atol=1e-14)
    assert np.allclose(K._block_to_node_displacement(u[2], cnv[2, 1]), K.block_to_node_kinematics(u, cnv)[2, 1])
    assert U.SolutionType.__args__ == (U.SolutionData, U.EigenmodeData)


def test_cross_correlation_measures():
    """objective.py:10-89 against the definition written out: sum_k a[k + lag] b[k] / max autocorrelation."""
    from difflexmm_amd import objective as O
    rng = np.random.default_rng(2)
    t = np.linspace(0, 1, 60)
    a = np.exp(-((t - 0.3) / 0.05) ** 2)
    b = 0.7 * np.exp(-((t - 0.45) / 0.05) ** 2)                      # the same pulse 9 samples later
    xc = O.compute_xcorr(a, b)
    assert xc.shape == (119,)
    lag = 4
    direct = sum(a[k + lag] * b[k] for k in range(60 - lag)) / (a * a).sum()
    assert abs(O.compute_xcorr(a, b, shift=lag) - direct) < 1e-14 and abs(O.compute_xcorr(a, a, shift=0) - 1.0) < 1e-14
    A = rng.normal(size=(5, 40))
    B = np.roll(A, 6, axis=1)                                        # every row delayed by 6 samples
    B[:, :6] = 0.0
    full = O.compute_xcorr2d(A, B)
    assert full.shape == (9, 79)
    i, j = 1, -3
    direct = sum(A[r + i, c + j] * B[r, c] for r in range(5) for c in range(40) if 0 <= r + i < 5 and 0 <= c + j < 40) / (A * A).sum()
    assert abs(O.compute_xcorr2d(A, B, shift=(i, j)) - direct) < 1e-13
    assert np.allclose(O.compute_xcorr2d(A, B, shift=(0, None)), full[4]) and np.allclose(O.compute_xcorr2d(A, B, shift=(None, 2)), full[:, 41])
    peak, delay = O.compute_space_time_xcorr(A, B)
    assert delay == 6 and 0.5 < peak <= 1.0
    assert O.compute_max_xcorr2d_at_shift(A, A, 0, shift_axis=1)[1] == 0


@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_native_design_map_equals_the_numpy_maps(cpu_lib, lattice):
    """dfx_design_forward / dfx_design_vjp (csrc/dfx_design.h: lattice map + polygon pass + compute_inertia + undeformed void angles fused in
    native code, and the cotangent of all of it) against the NumPy functions of difflexmm_amd/geometry.py they replace in the design loop
    (geometry.py:607-952, 71-160, 181-253 of the reference), 1e-13; a batch of designs in one call."""
    import math
    from difflexmm_amd import geometry as G
    if lattice == "quads":
        geo = G.QuadGeometry(7, 5, 15.0, 2.25)
        base = geo.get_design_from_rotated_square(25 * math.pi / 180)
    else:
        geo = G.KagomeGeometry(4, 3, 20.0 * np.array([[1.0, 0.0], [math.cos(math.pi / 3), math.sin(math.pi / 3)]]), 2.25)
        base = tuple(np.zeros(sh) for sh in geo.design_shapes())
    rng = np.random.default_rng(0)
    bonds, rho = geo.bond_connectivity(), 6.18e-9
    nm = G.NativeDesignMap(geo, bonds)
    designs = [tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base) for _ in range(3)]
    cen, cnv, inertia, va = nm.forward(cpu_lib, designs, rho)
    cb, gb, ib, vb = rng.normal(size=cnv.shape), rng.normal(size=cen.shape), rng.normal(size=inertia.shape) * 1e8, rng.normal(size=va.shape)
    mine = nm.vjp(cpu_lib, designs, rho, cb, gb, ib, vb)
    for i, d in enumerate(designs):
        c0, v0 = geo.geometry_from_design(*d)
        assert np.abs(cen[i] - c0).max() < 1e-13 and np.abs(cnv[i] - v0).max() < 1e-13
        assert np.abs(inertia[i] - G.compute_inertia(v0, rho)).max() < 1e-13 * np.abs(inertia[i]).max()
        assert np.abs(va[i] - G.void_angles0(v0, bonds)).max() < 1e-13
        ref = geo.vjp(d, cb[i] + G.void_angles0_vjp(v0, bonds, vb[i]), gb[i], props_bar=(rho * (ib[i][:, 0] + ib[i][:, 1]), rho * ib[i][:, 2]))
        for a, b in zip(mine[i], ref):
            assert np.abs(a - b).max() < 1e-13 * np.abs(b).max()
    # cotangents that are absent (None) are zeros
    only = nm.vjp(cpu_lib, designs, rho, cb)
    ref0 = geo.vjp(designs[0], cb[0])
    assert all(np.abs(a - b).max() < 1e-13 * np.abs(b).max() for a, b in zip(only[0], ref0))
