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


def test_struct_layouts_of_header_libraries_and_binding_agree(hip_lib, cpu_lib):
    """sizeof and every field offset of dfx_special / dfx_problem / dfx_params / dfx_grads / dfx_stats (the ControlParams tree of
    /root/reference/difflexmm/utils.py:48-163, flattened) as both libraries were compiled, against the hand-written ctypes mirrors:
    a reordered or resized field fails here (and at load time: declare() refuses such a library) instead of corrupting a solve."""
    import ctypes as C
    from difflexmm_amd import _binding as b
    want = b.abi_layout()
    assert len(want) == 53
    for lib in (hip_lib, cpu_lib):
        buf = (C.c_int32 * 64)()
        assert lib.dfx_abi_layout(buf, 64) == len(want)
        assert list(buf)[:len(want)] == want
    # the count the header documents, and a truncated query
    small = (C.c_int32 * 3)()
    assert hip_lib.dfx_abi_layout(small, 3) == 53 and list(small) == want[:3]
    # a mirror with a field out of place is caught
    class bad_stats(C.Structure):
        _fields_ = [("rhs_evals", C.c_int64), ("steps", C.c_int64)] + b.dfx_stats._fields_[2:]
    assert [getattr(bad_stats, n).offset for n, _ in bad_stats._fields_] == [getattr(b.dfx_stats, n).offset for n, _ in b.dfx_stats._fields_]   # same offsets, other names:
    assert [n for n, _ in bad_stats._fields_] != [n for n, _ in b.dfx_stats._fields_]                                                           # names are the header's job (test_header_and_binding_agree)


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


def test_dfx_library_cannot_point_at_the_cpu_port(cpu_lib, monkeypatch):
    """DFX_LIBRARY selects between builds of the gfx950 engine; pointing it at the CPU port of the oracle (same symbols) must fail
    loudly, and so must handing an unmarked non-gfx950 library to Engine / the problem classes."""
    import ctypes
    import numpy as np
    from difflexmm_amd import _binding as b
    cpu_so = os.path.join(ROOT, "oracle", "cpu", "libdfx_cpu.so")
    monkeypatch.setenv("DFX_LIBRARY", cpu_so)
    monkeypatch.setattr(b, "_LIB", None)
    with pytest.raises(RuntimeError, match="not a build of the gfx950 engine"):
        b.load_library()
    monkeypatch.delenv("DFX_LIBRARY")
    monkeypatch.setattr(b, "_LIB", None)
    raw = b.declare(ctypes.CDLL(cpu_so))                   # the same library WITHOUT the test-only mark oracle.cpu.load() sets
    with pytest.raises(RuntimeError, match="not the gfx950 engine"):
        b.Engine(4, 4, np.array([[0, 6]]), 1, 0, [], [], lib=raw)
    assert getattr(cpu_lib, "_dfx_test_only", False) is True


def test_test_library_hook_is_not_a_dataclass_field():
    import dataclasses
    from difflexmm_amd import problems as P
    for cls in (P.QuadsFocusingForward, P.KagomeFocusingForward, P.QuadsStaticTuningForward, P.QuadsSpinForward):
        assert "_lib" not in [f.name for f in dataclasses.fields(cls)]
