"""-m gpu: hinge characterisation (dynamics, reaction forces and their derivatives) on the HIP engine."""
import pytest

from . import hinge_common as HC

pytestmark = pytest.mark.gpu


def test_force_displacement_and_fit_gradient_hip(hip_lib):
    HC.check_force_displacement_and_fit_gradient(None)


def test_quads_sample_hip(hip_lib):
    HC.check_quads_sample(None)


def test_fit_loops_hip(hip_lib):
    HC.check_fit_loops(None)
