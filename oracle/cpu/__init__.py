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


def load_native():
    """The same port built ON THIS HOST with -march=native (the CPU baseline is timed with the host's own instruction set;
    the prebuilt library is x86-64-v3 so that it runs on whatever host the GPU box has).  Compiled into the temp directory on first use
    (~25 s), keyed by the sources and the CPU's flags.  Returns (handle, flags string); (None, reason) when no compiler is there."""
    import hashlib
    import tempfile
    from difflexmm_amd._binding import declare
    srcs = [os.path.join(_HERE, "dfx_cpu.cpp")] + [os.path.join(_HERE, "..", "..", "difflexmm_amd", "csrc", f)
                                                   for f in ("dfx_physics.h", "dfx_plan.h", "dfx_stage.h")] + [os.path.join(_HERE, "..", "..", "include", "dfx.h")]
    h = hashlib.sha256()
    for f in srcs:
        h.update(open(f, "rb").read())
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
        h.update(flags.encode())
    except Exception:       # noqa: BLE001
        pass
    # a directory only this user can write (round-4 advice: a predictable name in the shared temp directory could be pre-planted by
    # another local user and would then be loaded and timed), the intermediate file from mkstemp, ownership checked before loading
    cache = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "difflexmm_amd")
    try:
        os.makedirs(cache, mode=0o700, exist_ok=True)
        st = os.stat(cache)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise PermissionError(cache)
    except Exception:       # noqa: BLE001 -- no usable home: a fresh private directory for this process
        cache = tempfile.mkdtemp(prefix="dfx_cpu_native_")
    so = os.path.join(cache, f"libdfx_cpu_native_{h.hexdigest()[:12]}.so")
    cflags = "-O3 -march=native -std=c++17 -fPIC -fopenmp"
    if not os.path.exists(so):
        try:
            fd, tmp = tempfile.mkstemp(suffix=".so", dir=cache)
            os.close(fd)
            subprocess.check_call(["g++"] + cflags.split() + ["-Wno-unknown-pragmas", "-shared", "-o", tmp, srcs[0]],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, so)
        except Exception as e:       # noqa: BLE001
            return None, f"-march=native build failed ({type(e).__name__}): prebuilt x86-64-v3 library used"
    st = os.stat(so)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        return None, "cached -march=native library is not owned by this user: prebuilt x86-64-v3 library used"
    lib = declare(ctypes.CDLL(so))
    lib._dfx_test_only = True
    return lib, "g++ " + cflags + " (built on this host)"
