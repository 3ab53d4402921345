"""Hinge characterisation on the CPU port (tests/hinge_common.py; HIP twin: tests/test_gpu_hinge.py)."""
from . import hinge_common as HC


def test_force_displacement_and_fit_gradient_cpu_port(cpu_lib):
    HC.check_force_displacement_and_fit_gradient(cpu_lib)


def test_quads_sample_cpu_port(cpu_lib):
    HC.check_quads_sample(cpu_lib)


def test_fit_loops_cpu_port(cpu_lib):
    HC.check_fit_loops(cpu_lib)
