"""Long-horizon goldens (tests/golden/long_horizon_*.npz, made by make_golden.py long_horizon): 1200 / 800 Dopri5 steps of the torch
oracle kept on the autograd tape -- trajectory at 9 output times, target kinetic energy and its design gradient -- against an engine
library (the HIP engine in the -m gpu test, the C++ port in the CPU test: that pins the second checker to the oracle over a horizon
the live oracle tests cannot afford)."""
import os

import numpy as np

from .common import Case, relerr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# stated tolerances: rounding differences between the engine's hand-derived forces (fused multiply-adds, its own summation order) and
# torch's autograd grow ~100x per 400 steps of these contact-engaged trajectories (make_golden.py long_horizon has the measured growth)
TOL_FIELDS, TOL_OBJECTIVE, TOL_GRAD = 1e-9, 1e-9, 1e-8


def check(lib, lattice, tol_fields=TOL_FIELDS, tol_objective=TOL_OBJECTIVE, tol_grad=TOL_GRAD):
    """`lattice`: "quads" (8 x 8, 1 200 steps), "kagome" (4 x 4 cells, 800 steps), or "quads32": 32 x 32 quads, 2 000 steps -- a size
    at which the engine's size-dependent code runs (several workgroups, XCD-banded order, more than one segment, the persistent loop's
    ring wrapping 1 500 times); its gradient was assembled interval by interval (make_golden.py long_horizon_32) and every other
    output row travels; "quads128": BASELINE config 3's lattice at full size, 128 x 128 quads, 500 contact-engaged steps in 25 intervals
    (make_golden.py long_horizon_128): full-size long-horizon parity against the TORCH oracle, not against the port that shares the
    engine's physics headers."""
    g = np.load(os.path.join(GOLD, f"long_horizon_{lattice}.npz"))
    lattice = "quads" if lattice.startswith("quads") else lattice
    c = Case(lattice, int(g["n"]), True, True, seed=int(g["seed"]), lib=lib, cutoff_deg=float(g["cutoff_deg"]))
    design = tuple(g[f"design_{i}"] for i in range(len(c.design)))
    for a, b in zip(design, c.design):
        assert np.array_equal(a, b)                    # the seeded case is the one the golden was made from
    assert g["contact_energy"].min() > 0.0             # contact engaged at every output time
    cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=3000.0, input_delay=1e-5))
    nb = c.geo.n_blocks
    f = c.solver(np.zeros((2, nb, 3)), g["timepoints"], cp, keep_trajectory=True, steps_per_interval=int(g["spi"]))
    assert c.solver.stats["steps"] == int(g["spi"]) * (len(g["timepoints"]) - 1)
    row_tol = None
    if "rows" in g.files:
        f = f[g["rows"]]
        # the golden carries how far the oracle's own two paths (taped interval by interval / tape-free) have drifted apart at every output:
        # this contact-engaged trajectory amplifies a last-bit difference ~10 x per 250 steps (1.6e-15 after 250 steps, 2.4e-8 after 2 000),
        # so the bound on the FIELDS follows the row -- 50 x that drift, at least 1e-11 -- while objective and gradient keep the flat bounds
        row_tol = np.maximum(1e-11, 50.0 * g["oracle_paths_agree"][g["rows"]])
    out = dict(q=relerr(f[:, 0], g["fields"][:, 0]), v=relerr(f[:, 1], g["fields"][:, 1]))
    obj, tree, _ = c.solver.kinetic_energy_value_and_vjp(g["target"].astype(np.int32))
    grads = c.geo.vjp(design, tree.geometrical_params.centroid_node_vectors, tree.geometrical_params.block_centroids)
    out["objective"] = abs(obj - float(g["objective"])) / abs(float(g["objective"]))
    out["grad"] = max(relerr(a, g[f"grad_{i}"]) for i, a in enumerate(grads))
    if row_tol is not None:
        for i in range(1, len(row_tol)):
            rq, rv = relerr(f[i, 0], g["fields"][i, 0]), relerr(f[i, 1], g["fields"][i, 1])
            out[f"row{int(g['rows'][i])}"] = (rq, rv, float(row_tol[i]))
            assert rq < row_tol[i] and rv < row_tol[i], out
    else:
        assert out["q"] < tol_fields and out["v"] < tol_fields, out
    assert out["objective"] < tol_objective, out
    assert out["grad"] < tol_grad, out
    return out
