"""Static-tuning drive + forward problem + weighted objective (problems/quads_kinetic_energy_static_tuning.py:124-283, 453-478)
against the oracle twin (oracle/ref_problems.py), on the CPU port of the engine: host logic, time functions, chain rules."""
from . import static_tuning_common as S


def test_boundary_conditions_and_drive_equal_oracle(cpu_lib):
    S.check_boundary_conditions_equal_oracle(cpu_lib)


def test_trajectory_objective_and_gradients_vs_autograd(cpu_lib):
    S.check_trajectory_and_gradients(cpu_lib)


def test_members_on_their_own_time_grids_equal_separate_calls(cpu_lib):
    S.check_members_on_their_own_grids_equal_separate_calls(cpu_lib)


def test_rows_with_equal_time_grids_are_ensemble_members(cpu_lib):
    S.check_rows_with_equal_grids_share_one_call(cpu_lib)


def test_optimisation_loop_and_best_forwards(cpu_lib):
    S.check_optimisation_loop(cpu_lib)


def test_spin_problem_angular_momentum_vs_autograd(cpu_lib):
    """problems/quads_spin.py on the CPU port: harmonic drive, angular-momentum objective and design gradient vs the oracle twin."""
    from . import spin_common
    spin_common.check_angular_momentum_value_and_gradient(cpu_lib)
