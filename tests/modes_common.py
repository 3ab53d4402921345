"""`linear_mode_analysis` (dynamics.py:189-245) on the engine -- the stiffness matrix assembled from the device's Hessian-vector hook --
against the oracle's autograd Hessian + the same generalised eigen-solve."""
import numpy as np

from difflexmm_amd.dynamics import linear_mode_analysis
from oracle import ref_dynamics as OD

from .common import Case, relerr


def check(lib, lattice, n, contact):
    cut = 125.0 if lattice == "kagome" else 42.0
    c = Case(lattice, n, True, contact, seed=17, lib=lib, cutoff_deg=cut, per_bond_k=True)
    u = 0.3 * c.random_state()[0]                       # linearise around a deformed configuration (contact engaged when on)
    w2, modes, K = linear_mode_analysis(u, c.geo, c.energy, c.cp, c.con, _lib=lib, return_stiffness=True)
    ow2, omodes, oK = OD.linear_mode_analysis(u, c.ogeo, c.oenergy, c.oracle_cp(), c.con)
    nf = 3 * c.geo.n_blocks - len(c.con)
    assert K.shape == (nf, nf) and modes.shape == (nf, c.geo.n_blocks, 3)
    assert relerr(K, oK) < 1e-11
    assert relerr(w2, ow2) < 1e-9
    con = np.asarray(c.con)
    assert np.all(modes[:, con[:, 0], con[:, 1]] == 0.0)
    assert np.allclose(np.linalg.norm(modes.reshape(nf, -1), axis=1), 1.0, atol=1e-12)
    # well-separated modes agree up to their sign
    gaps = np.minimum(np.diff(w2, prepend=-np.inf), np.diff(w2, append=np.inf))
    sep = np.where(gaps > 1e-6 * np.abs(w2).max())[0]
    assert len(sep) > nf // 2
    dots = np.abs(np.einsum("ij,ij->i", modes.reshape(nf, -1)[sep], omodes.reshape(nf, -1)[sep]))
    assert np.all(dots > 1 - 1e-6), dots.min()
    return w2
