"""distance-based contact (difflexmm/energy.py:222-330, build_contact_energy(angle_based=False) :364-407) through the
engine: energy, one RHS + every VJP (node vectors, block centroids, contact constants, inertia, state) against autograd through
the oracle's restatement, a trajectory + discrete adjoint w.r.t. the design against autograd through the unrolled oracle solver,
and the host NumPy restatement in difflexmm_amd.energy.  CPU port here, HIP in the -m gpu twins."""
import math

import numpy as np
import pytest
import torch

import difflexmm_amd as dm
from difflexmm_amd import energy as en_mod
from difflexmm_amd import geometry as geo_mod
from difflexmm_amd import loading as ld
from difflexmm_amd.dynamics import setup_dynamic_solver
from oracle import ref_dynamics as OD
from oracle import ref_energy as OE
from oracle import ref_geometry as OG

from .common import DENSITY, K_ROT, K_SHEAR, K_STRETCH, relerr, torch_pulse

T64 = lambda x, g=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=g)  # noqa: E731


class DistCase:
    """Quads or kagome with nonlinear ligaments + distance contact; thresholds chosen so that a good part of the void-edge
    distances is inside [min, cutoff) (the rest exercise the inactive branches)."""

    def __init__(self, lattice, lib, n=4, seed=3, batch=1):
        rng = self.rng = np.random.default_rng(seed)
        if lattice == "quads":
            self.geo, self.ogeo = geo_mod.QuadGeometry(n, n, 15.0, 2.25), OG.QuadGeometry(n, n, 15.0, 2.25)
            base = self.geo.get_design_from_rotated_square(25 * math.pi / 180)
            self.design = tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base)
            mid = (n // 2) * n
        else:
            basis = 20.0 * np.array([[1.0, 0.0], [math.cos(math.pi / 3), math.sin(math.pi / 3)]])
            self.geo, self.ogeo = geo_mod.KagomeGeometry(n, n, basis, 2.25), OG.KagomeGeometry(n, n, basis, 2.25)
            self.design = tuple(rng.uniform(-0.3, 0.3, s) for s in self.geo.design_shapes())
            mid = 2 * n * (n // 2)
        self.bonds = self.geo.bond_connectivity()
        self.cnv, self.cen = self.geo.centroid_node_vectors(*self.design), self.geo.block_centroids(*self.design)
        self.refv = self.geo.reference_bond_vectors()
        self.con = np.array([[mid, 0], [mid, 1], [mid, 2], [0, 0], [0, 1], [0, 2]])
        self.vec = np.array([1.0, 0, 0, 0, 0, 0])
        d0 = en_mod.build_void_edge_distance(self.bonds)(self.cen[:, None] + self.cnv)
        # at rest every void-edge distance is the hinge-to-hinge distance = bond length (2.25): the penalty is active on all of them
        assert abs(np.median(d0) - 2.25) < 1e-9
        self.contact_params = (1.0, 2.6, 0.5)                                      # (min, cutoff, k): lengths
        energy = en_mod.combine_block_energies(en_mod.build_strain_energy(self.bonds, en_mod.ligament_energy),
                                               en_mod.build_contact_energy(self.bonds, angle_based=False))
        self.solver = setup_dynamic_solver(self.geo, energy, constrained_block_DOF_pairs=self.con, constrained_DOFs_fn=ld.Pulse(self.vec),
                                           batch=batch, _lib=lib)
        self.oenergy = OE.combine_block_energies(OE.build_strain_energy(self.bonds, OE.ligament_energy),
                                                 OE.build_contact_energy(self.bonds, angle_based=False))
        self.pulse = dict(amplitude=1.5, loading_rate=3000.0, input_delay=1e-5)     # moderate: the hinges must not collapse onto the asymptote
        self.cp = dm.ControlParams(dm.GeometricalParams(self.cen, self.cnv),
                                   dm.MechanicalParams(dm.LigamentParams(K_STRETCH, K_SHEAR, K_ROT, self.refv), DENSITY, None, 0.0,
                                                       dm.ContactParams(*self.contact_params)),
                                   constraint_params=dict(self.pulse))

    def oracle_cp(self, cnv=None, cen=None, contact=None, amplitude=None):
        p = {k: T64(v) for k, v in self.pulse.items()}
        if amplitude is not None:
            p["amplitude"] = amplitude
        cpar = [T64(v) for v in self.contact_params] if contact is None else contact
        return OE.ControlParams(OE.GeometricalParams(T64(self.cen) if cen is None else cen, T64(self.cnv) if cnv is None else cnv),
                                OE.MechanicalParams(OE.LigamentParams(T64(K_STRETCH), T64(K_SHEAR), T64(K_ROT), T64(self.refv)), T64(DENSITY), None,
                                                    T64(0.0), OE.ContactParams(*cpar)), constraint_params=p)

    def oracle_solver(self, **kw):
        return OD.setup_dynamic_solver(self.ogeo, self.oenergy, constrained_block_DOF_pairs=self.con,
                                       constrained_DOFs_fn=torch_pulse(self.vec), **kw)


def test_host_restatement_of_the_void_edge_distances_equals_the_oracle():
    geo = geo_mod.KagomeGeometry(3, 3, 20.0 * np.array([[1.0, 0.0], [0.5, math.sqrt(3) / 2]]), 2.25)
    rng = np.random.default_rng(0)
    design = tuple(rng.uniform(-0.3, 0.3, s) for s in geo.design_shapes())
    nodes = geo.block_centroids(*design)[:, None] + geo.centroid_node_vectors(*design) + rng.normal(size=(geo.n_blocks, 3, 2)) * 0.2
    bonds = geo.bond_connectivity()
    mine = en_mod.build_void_edge_distance(bonds)(nodes)
    ref = OE.build_void_edge_distance(bonds)(T64(nodes)).numpy()
    assert mine.shape == (2 * len(bonds),) and np.abs(mine - ref).max() < 1e-13 * np.abs(ref).max()


def check_rhs_and_vjp(lib, lattice):
    c = DistCase(lattice, lib, n=4 if lattice == "quads" else 3)
    s = c.solver
    flat = s._flatten(c.cp)
    assert "block_centroids" in flat and "void_angle0" not in flat
    s.engine.set_params(**{k: v[None] for k, v in flat.items()})
    y = c.rng.normal(size=(2, c.geo.n_blocks, 3)) * np.array([0.3, 0.3, 0.1])
    y[1] *= 50.0
    lam = c.rng.normal(size=y.shape)
    osol = c.oracle_solver()
    free = osol.free_DOF_ids
    cnv, cen, inertia = T64(c.cnv, True), T64(c.cen, True), T64(flat["inertia"], True)
    contact = [T64(v, True) for v in c.contact_params]
    yf = T64(y.reshape(2, -1)[:, free], True)
    # energy of the configuration (constraints applied through the kinematics): engine vs oracle
    u_full = osol.kinematics(yf[0], 0.012, c.oracle_cp().constraint_params)
    e_oracle = float(c.oenergy(u_full.detach(), c.oracle_cp()))
    e_engine = s.engine.energy(u_full.detach().numpy().reshape(1, -1, 3))[0]
    e_strain = float(OE.build_strain_energy(c.bonds, OE.ligament_energy)(u_full.detach(), c.oracle_cp()))
    assert e_oracle - e_strain > 1e-6 * e_oracle, "contact inactive: the test would be vacuous"
    assert abs(e_engine - e_oracle) < 1e-11 * e_oracle
    dy = s.engine.rhs(y[None], 0.012)[0]
    yb, g = s.engine.rhs_vjp(y[None], 0.012, lam[None])
    r = osol.rhs(yf, 0.012, c.oracle_cp(cnv=cnv, cen=cen, contact=contact), inertia.reshape(-1)[torch.as_tensor(free)], create_graph=True)
    L = (r * T64(lam.reshape(2, -1)[:, free])).sum()
    gr = torch.autograd.grad(L, [yf, cnv, cen, inertia] + contact)
    assert relerr(dy.reshape(2, -1)[:, free], r.detach().numpy()) < 1e-11
    assert relerr(yb[0].reshape(2, -1)[:, free], gr[0].numpy()) < 1e-10
    assert relerr(g["centroid_node_vectors"][0], gr[1].numpy()) < 1e-10
    assert relerr(g["block_centroids"][0], gr[2].numpy()) < 1e-10
    assert relerr(g["inertia"][0], gr[3].numpy()) < 1e-11
    assert relerr(g["contact"][0], np.array([x.item() for x in gr[4:]])) < 1e-10


def check_trajectory_and_adjoint(lib, lattice, spi=6, n_out=4):
    c = DistCase(lattice, lib, n=4 if lattice == "quads" else 3, seed=5)
    ts = np.linspace(0, 3e-4, n_out)
    y0 = c.rng.normal(size=(2, c.geo.n_blocks, 3)) * np.array([0.05, 0.05, 0.02])
    y0[1] *= 5.0
    fields = c.solver(y0, ts, c.cp, keep_trajectory=True, steps_per_interval=spi)
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=spi)
    # (the penalty is stiff near its asymptote and the reference's sqrt(|a|^2 - t^2 |e|^2) loses digits the closest-point form
    # keeps: 1e-9 here instead of the 1e-10 of the other models)
    assert relerr(fields, osol(y0, ts, c.oracle_cp()).numpy()) < 1e-9
    fb = c.rng.normal(size=fields.shape)
    fb.reshape(len(ts), 2, -1)[:, :, c.solver.constrained_DOF_ids] = 0.0
    tree, s0 = c.solver.vjp(fb)
    design = [T64(d, True) for d in c.design]
    amp, y0t = T64(1.5, True), T64(y0, True)
    hist, _ = OD.solve_fixed_differentiable(osol, c.ogeo, y0t, ts, c.oracle_cp(cnv=c.ogeo.centroid_node_vectors(*design),
                                                                                 cen=c.ogeo.block_centroids(*design), amplitude=amp), spi)
    free = osol.free_DOF_ids
    L = (hist * T64(fb.reshape(len(ts), 2, -1)[:, :, free])).sum()
    gr = torch.autograd.grad(L, design + [amp, y0t])
    assert np.abs(tree.geometrical_params.block_centroids).max() > 0          # the centroids matter in this contact model
    mine = c.geo.vjp(c.design, tree.geometrical_params.centroid_node_vectors, tree.geometrical_params.block_centroids)
    for a, b in zip(mine, gr[:len(design)]):
        assert relerr(a, b.numpy()) < 1e-9
    assert abs(tree.constraint_params["amplitude"] - gr[len(design)].item()) / abs(gr[len(design)].item()) < 1e-9
    assert relerr(s0.reshape(2, -1)[:, free], gr[-1].numpy().reshape(2, -1)[:, free]) < 1e-9


@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_distance_contact_rhs_and_vjp_cpu_port(cpu_lib, lattice):
    check_rhs_and_vjp(cpu_lib, lattice)


@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_distance_contact_trajectory_and_adjoint_cpu_port(cpu_lib, lattice):
    check_trajectory_and_adjoint(cpu_lib, lattice)


@pytest.mark.gpu
@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_distance_contact_rhs_and_vjp_hip(hip_lib, lattice):
    check_rhs_and_vjp(None, lattice)


@pytest.mark.gpu
@pytest.mark.parametrize("lattice", ["quads", "kagome"])
def test_distance_contact_trajectory_and_adjoint_hip(hip_lib, lattice):
    check_trajectory_and_adjoint(None, lattice)
