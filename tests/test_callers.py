"""The remaining callers of the hot path on the CPU port (tests/callers_common.py; HIP twin: tests/test_gpu_callers.py)."""
from . import callers_common as C


def test_constraints_and_jacobians_equal_the_oracle_twin():
    C.check_constraints_against_oracle()


def test_energy_splitting_objective_cpu_port(cpu_lib):
    C.check_energy_splitting(cpu_lib)


def test_restricted_design_space_cpu_port(cpu_lib):
    C.check_restricted_design_space(cpu_lib)


def test_reference_design_cpu_port(cpu_lib):
    C.check_reference_design(cpu_lib)


def test_recorded_input_signal_cpu_port(cpu_lib):
    C.check_recorded_input_signal(cpu_lib)


def test_more_designs_than_batch_cpu_port(cpu_lib):
    C.check_more_designs_than_batch(cpu_lib)
