"""Loader of the CPU port (oracle/cpu/libdfx_cpu.so) -- test infrastructure / CPU baseline only."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    # DFX_CPU_PORT_LIBRARY: another build of the same port (the sanitizer build of `make asan`, tools/sanitize_cpu.sh)
    if os.environ.get("DFX_CPU_PORT_LIBRARY"):
        return os.environ["DFX_CPU_PORT_LIBRARY"]
    so = os.path.join(_HERE, "libdfx_cpu.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return so


def load():
    """ctypes handle with the same ABI as libdfx (include/dfx.h), argument types declared."""
    global _LIB
    if _LIB is None:
        from difflexmm_amd._binding import declare
        _LIB = declare(ctypes.CDLL(build()))
        _LIB._dfx_test_only = True        # the product's Engine refuses any non-gfx950 library without this mark
        # hosts report hundreds of logical CPUs but may grant only a few: a spinning 256-thread OpenMP team on small
        # lattices is pathologically slow, so cap the team unless the user asked for something else
        if "OMP_NUM_THREADS" not in os.environ:
            set_threads(min(os.cpu_count() or 1, 8))
    return _LIB


def set_threads(n):
    """Number of OpenMP threads of the CPU port (libgomp)."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
