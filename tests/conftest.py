import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cpu_lib():
    """The CPU port of the engine (oracle/cpu) -- used by the no-GPU suite to exercise host logic + physics."""
    from oracle.cpu import load
    return load()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; loading it without a GPU is fine, creating a solver is not."""
    from difflexmm_amd._binding import load_library
    return load_library()
