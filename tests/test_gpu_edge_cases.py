"""-m gpu: the edge cases of tests/test_edge_cases.py on the HIP engine."""
import numpy as np
import pytest

from .test_edge_cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_hip(hip_lib, name):
    CASES[name](None)



def test_a_diverging_member_is_flagged_not_fatal(hip_lib):
    """Failure isolation on the device (dfx_member_status / dfx_set_failure_policy): three members of a batch, the middle one
    with a block of almost no mass -- its state overflows within a few steps.  Default policy: the call fails and names the member.
    Isolating policy: the call returns, member 1 is NaN and flagged, members 0 and 2 are bit-identical to a batch without the bad
    member's blow-up (they never see it), and the reverse sweep leaves their gradients untouched.  Fixed grid (the state overflows:
    status 1) and the adaptive controller, stage launches and the persistent loop (the controller follows the stiff member with ever
    smaller steps until its budget is spent: status 3; jax's odeint has no budget -- mxstep = inf -- and would not return)."""
    from tests.test_multiprocess import _collapsed, _make
    import os
    fw, obj, designs = _make(None)
    eng_args = dict(batch=3)
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    import math
    def problem(spi):
        f = QuadsFocusingForward(n1_blocks=6, n2_blocks=6, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9,
                                 damping=1e-4 * np.ones((36, 3)), amplitude=7.5, loading_rate=3000.0, input_delay=1e-5, n_excited_blocks=2,
                                 loaded_side="left", input_shift=0, simulation_time=4e-4, n_timepoints=5, use_contact=True, k_contact=1.5,
                                 min_angle=-15 * math.pi / 180, cutoff_angle=42 * math.pi / 180, steps_per_interval=spi, batch=3, rtol=1e-7, atol=1e-7)
        return f, TargetKineticEnergy(f, (2, 2), (1, 1))
    good = [designs[0], designs[1], designs[2]]
    bad = [designs[0], _collapsed(fw, designs[1]), designs[2]]
    for spi, env in ((10, {}), (None, {"DFX_PERSIST": "0"}), (None, {"DFX_PERSIST": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            f, o = problem(spi)
            f.solve_dynamics.max_attempts = 1500
            v_ref, g_ref = o.value_and_grad(good)
            eng = f.solve_dynamics.engine
            assert eng.member_status().tolist() == [0, 0, 0]
            with pytest.raises(RuntimeError, match="member 1"):
                o.value_and_grad(bad)
            eng.set_failure_policy(True)
            v, g = o.value_and_grad(bad)
            assert eng.member_status().tolist() == [0, 1 if spi else 3, 0], (spi, env, eng.member_status())
            assert np.isnan(v[1]) and v[0] == v_ref[0] and v[2] == v_ref[2], (spi, env, v, v_ref)
            for m in (0, 2):
                for a, b in zip(g[m], g_ref[m]):
                    assert np.array_equal(a, b)
            eng.close()
        finally:
            for k, val in old.items():
                os.environ.pop(k, None) if val is None else os.environ.__setitem__(k, val)


def test_state0_must_be_finite(hip_lib):
    from tests.common import Case
    c = Case("quads", 4, True, False, seed=1)
    y0 = np.zeros((2, 16, 3)); y0[0, 3, 1] = np.inf
    with pytest.raises(RuntimeError, match="non-finite"):
        c.solver(y0, np.linspace(0, 1e-4, 3), c.cp, steps_per_interval=4)
    with pytest.raises(RuntimeError, match="non-finite"):
        c.solver(y0, np.linspace(0, 1e-4, 3), c.cp)
