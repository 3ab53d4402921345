"""The C-ABI library loads and exports every symbol include/dfx.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dfx.h")).read()
    return sorted(set(re.findall(r"\b(dfx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from difflexmm_amd._binding import EXPORTS
    assert sorted(EXPORTS) == declared_symbols()


def test_libdfx_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), name
    assert b"gfx950" in hip_lib.dfx_version()


def test_cpu_port_exports_the_solver_abi(cpu_lib):
    """The CPU port (test infrastructure) mirrors the solver entry points; the RCCL collective and the device helpers exist in
    the HIP library only."""
    from difflexmm_amd._binding import COMM_EXPORTS
    for name in declared_symbols():
        if name not in COMM_EXPORTS:
            assert hasattr(cpu_lib, name), name


def test_no_pytorch_in_the_product():
    """north_star: host = Python + ctypes.  Nothing under difflexmm_amd/ and nothing in bench.py imports torch."""
    import re
    paths = [os.path.join(ROOT, "bench.py")]
    for dirpath, _, files in os.walk(os.path.join(ROOT, "difflexmm_amd")):
        paths += [os.path.join(dirpath, f) for f in files if f.endswith(".py")]
    for path in paths:
        assert not re.search(r"^\s*(from|import)\s+torch\b", open(path).read(), re.M), path


def test_product_has_no_cpu_fallback(hip_lib):
    """Without a HIP device the product must fail loudly instead of computing somewhere else."""
    import numpy as np
    if hip_lib.dfx_device_count() > 0:
        pytest.skip("a GPU is present")
    from difflexmm_amd import _binding as b
    with pytest.raises(RuntimeError, match="no HIP device"):
        b.Engine(4, 4, np.array([[0, 6]]), 1, 0, [], [])


def test_product_never_imports_the_oracle():
    """Nothing under difflexmm_amd/ imports, includes or loads anything from oracle/."""
    import re
    pkg = os.path.join(ROOT, "difflexmm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                src = open(path).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), path
                assert "libdfx_cpu" not in src, path
            elif f.endswith((".h", ".hip", ".cpp")):
                src = open(path).read()
                assert not re.search(r'#include\s+"[^"]*oracle/', src), path
