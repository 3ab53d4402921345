"""-m gpu: energy splitting, restricted design space and the rotated-squares reference design on the HIP engine."""
import pytest

from . import callers_common as C

pytestmark = pytest.mark.gpu


def test_energy_splitting_objective_hip(hip_lib):
    C.check_energy_splitting(None)


def test_restricted_design_space_hip(hip_lib):
    C.check_restricted_design_space(None)


def test_reference_design_hip(hip_lib):
    C.check_reference_design(None)


def test_recorded_input_signal_hip(hip_lib):
    C.check_recorded_input_signal(None)


def test_more_designs_than_batch_hip(hip_lib):
    C.check_more_designs_than_batch(None)
