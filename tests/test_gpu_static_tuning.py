"""-m gpu: the static-tuning problem on the HIP engine against the torch oracle (trajectory 1e-10, gradients w.r.t. design,
amplitude, loading rate, compressive strain and strain rate 1e-9)."""
import pytest

from . import static_tuning_common as S

pytestmark = pytest.mark.gpu


def test_boundary_conditions_and_drive_equal_oracle(hip_lib):
    S.check_boundary_conditions_equal_oracle(None)


def test_trajectory_objective_and_gradients_vs_autograd(hip_lib):
    S.check_trajectory_and_gradients(None)


def test_members_on_their_own_time_grids_equal_separate_calls(hip_lib):
    S.check_members_on_their_own_grids_equal_separate_calls(None)


def test_rows_with_equal_time_grids_are_ensemble_members(hip_lib):
    S.check_rows_with_equal_grids_share_one_call(None)


def test_optimisation_loop_and_best_forwards(hip_lib):
    S.check_optimisation_loop(None)


def test_spin_problem_angular_momentum_vs_autograd(hip_lib):
    """problems/quads_spin.py on the HIP engine: harmonic drive, angular-momentum objective and design gradient vs the oracle twin."""
    from . import spin_common
    spin_common.check_angular_momentum_value_and_gradient(None)


@pytest.mark.parametrize("level", ["records", "stages", "state", "segments"])
def test_members_on_their_own_time_grids_at_every_checkpoint_level(hip_lib, monkeypatch, level):
    """Per-member time grids (dfx_forward_grid_members) through every reverse-sweep variant: the records build, the rebuild build
    (stages: k_rebuild_first reads the member's own last step), the recompute chains (state) and the interval re-runs (segments:
    k_init reads the member's own interval start)."""
    monkeypatch.setenv("DFX_CHECKPOINT", level)
    S.check_members_on_their_own_grids_equal_separate_calls(None)
