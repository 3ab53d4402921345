"""Generates tests/golden/*.npz from the oracle (oracle/ref_*.py: torch-fp64 + autograd restatement of the reference).

The reference itself cannot be imported in the build container (JAX absent) and ships no golden arrays, so these
fixtures are outputs of the oracle restatement on seeded inputs; they freeze it (any later edit of the oracle or of
the engine shows up as a diff) and travel to the GPU box.  Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_dynamics as OD  # noqa: E402
from oracle.cpu import load  # noqa: E402
from tests.common import Case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T64 = lambda x, g=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=g)  # noqa: E731


def rhs_cases():
    out = {}
    for lattice, n, cut in (("quads", 4, 42.0), ("kagome", 3, 125.0)):
        for nonlinear in (True, False):
            c = Case(lattice, n, nonlinear, True, seed=7, lib=load(), cutoff_deg=cut)
            y = c.random_state()
            osol = c.oracle_solver()
            free = osol.free_DOF_ids
            r = osol.rhs(T64(y.reshape(2, -1)[:, free]), 0.012, c.oracle_cp(), osol.reduced_inertia(c.oracle_cp()))
            key = f"{lattice}{n}_{'nl' if nonlinear else 'lin'}"
            out[key + "_y"] = y
            out[key + "_dy_free"] = r.detach().numpy()
            out[key + "_free"] = free
    np.savez_compressed(os.path.join(OUT, "rhs.npz"), **out)


def adaptive_trajectory():
    """C1-like: 8x8 quads, linearised ligaments, no contact, no damping; adaptive Dopri5 (reference odeint semantics)."""
    c = Case("quads", 8, False, False, damping=False, seed=1, lib=load())
    fast = dict(loading_rate=T64(600.0), input_delay=T64(1e-4))
    ts = np.linspace(0.0, 4e-3, 11)
    osol = c.oracle_solver(rtol=1e-8, atol=1e-8)
    f = osol(np.zeros((2, 64, 3)), ts, c.oracle_cp(fast)).numpy()
    np.savez_compressed(os.path.join(OUT, "adaptive_8x8.npz"), timepoints=ts, fields=f, attempted=osol.stats["attempted"],
                        accepted=osol.stats["accepted"], loading_rate=600.0, input_delay=1e-4)


def focusing_gradient():
    """6x6 quads focusing problem: objective = target kinetic energy, gradient w.r.t. the design by autograd through the
    unrolled fixed-grid oracle (the oracle's stand-in for jax.grad through solve_dynamics)."""
    from oracle import ref_energy as OE, ref_geometry as OG
    c = Case("quads", 6, True, True, seed=9, lib=load(), cutoff_deg=42.0)
    lv = dict(loading_rate=T64(3000.0), input_delay=T64(1e-5))
    ts = np.linspace(0.0, 4e-4, 5)
    spi = 8
    design = [T64(d, True) for d in c.design]
    cnv = c.ogeo.centroid_node_vectors(*design)
    cen = c.ogeo.block_centroids(*design)
    osol = c.oracle_solver(integrator="fixed", steps_per_interval=spi)
    hist, _ = OD.solve_fixed_differentiable(osol, c.ogeo, T64(np.zeros((2, 36, 3))), ts, c.oracle_cp(dict(cnv=cnv, cen=cen, **lv)), spi)
    target = np.array([14, 15, 20, 21])
    free = list(osol.free_DOF_ids)
    inertia = OG.compute_inertia(cnv, 6.18e-9)
    obj = 0.0
    for b in target:
        for d in range(3):
            col = free.index(b * 3 + d)
            obj = obj + (inertia[b, d] * hist[:, 1, col] ** 2 / 2).sum()
    g = torch.autograd.grad(obj, design)
    np.savez_compressed(os.path.join(OUT, "focusing_6x6.npz"), timepoints=ts, spi=spi, target=target, objective=obj.item(),
                        grad_h=g[0].numpy(), grad_v=g[1].numpy(), design_h=c.design[0], design_v=c.design[1])


if __name__ == "__main__":
    rhs_cases()
    adaptive_trajectory()
    focusing_gradient()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
