"""The opt-in experiments' tests (two stages per launch on lattice windows, every ligament once on lattice tiles): run with
    make -C difflexmm_amd/csrc experimental
    DFX_LIBRARY=difflexmm_amd/libdfx_experimental.so python -m pytest variants/experimental/tests -m gpu
They use the fixtures of the main suite (tests/conftest.py) and skip unless the library says it is an experimental build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from tests.conftest import cpu_lib, experimental_lib, hip_lib, pytest_configure, pytest_sessionstart  # noqa: E402,F401
