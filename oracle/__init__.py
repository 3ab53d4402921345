"""CPU oracle for the DifFlexMM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-fp64 tensors for the energy functionals so that
``torch.autograd`` plays the role ``jax.grad`` plays in the reference, NumPy for the
integrator) of the reference's per-timestep force assembly and ODE integration:

* ``ref_geometry``  follows ``difflexmm/geometry.py``
* ``ref_energy``    follows ``difflexmm/energy.py`` and ``difflexmm/kinematics.py``
* ``ref_dynamics``  follows ``difflexmm/dynamics.py`` and ``difflexmm/loading.py``
* ``ref_ode``       restates ``jax.experimental.ode.odeint`` (jax 0.4.8, third party, not
                    vendored in the reference; pinned in ``poetry.lock:614-615``)
* ``cpu/``          a C++17 restatement (hand-derived forces) used as the timed CPU baseline

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from here.  The product (``difflexmm_amd``) never does and fails loudly when
its HIP library is missing.

PARITY PINNING.  JAX / jax-md are not importable in the build container and the reference
ships no golden arrays, so the oracle cannot be compared with outputs of the reference
itself.  It is pinned by everything the reference's own tests hold for this path:

* ``tests/test_difflexmm.py:149-176`` (rigid-rotation frame invariance, E < 1e-30), and
* ``tests/test_difflexmm.py:35-146``  (tensile known-answer test, 1e-4 relative),
* the inertia constants hard-coded in the reference notebooks (0.36125, 0.0217502604),

all reproduced in ``tests/test_oracle_kat.py``.  Contact energy, displacement driving, velocity
reconstruction and every gradient are "parity unpinned" by the reference's tests; for those the
oracle is a line-by-line restatement cross-checked by finite differences and SciPy.  The
Dormand-Prince constants recalled from jax.experimental.ode are pinned against SciPy's RK45
(same published tableau: exact match of nodes, stage coefficients, weights and mid-point
interpolation; the embedded error weights are Shampine's variant, -2/3 of SciPy's).
"""
