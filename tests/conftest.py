import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared libraries are build artefacts (git-ignored): in a fresh checkout build them the way
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; `make` is a no-op when they are current)."""
    import shutil
    import subprocess
    need_hip = not os.path.exists(os.path.join(ROOT, "difflexmm_amd", "libdfx.so"))
    need_cpu = not os.path.exists(os.path.join(ROOT, "oracle", "cpu", "libdfx_cpu.so"))
    if need_hip and shutil.which("hipcc") or need_hip and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "difflexmm_amd", "csrc")], stdout=subprocess.DEVNULL)
    if need_cpu:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "cpu")], stdout=subprocess.DEVNULL)


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


@pytest.fixture(scope="session")
def experimental_lib(hip_lib):
    """The opt-in experiments (two stages per launch on lattice windows, every ligament once on lattice tiles) and the test hook
    DFX_TEST_FREE_BYTES are compiled into `make -C difflexmm_amd/csrc experimental` only (libdfx_experimental.so), never into the
    default library: their tests run when DFX_LIBRARY points at such a build and skip otherwise."""
    if b"experimental" not in hip_lib.dfx_version():
        pytest.skip("needs a library built with -DDFX_EXPERIMENTAL (make -C difflexmm_amd/csrc experimental; DFX_LIBRARY=difflexmm_amd/libdfx_experimental.so)")
    return hip_lib
