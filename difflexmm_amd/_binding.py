"""ctypes binding of the libdfx C ABI (``include/dfx.h``).

The product loads ``difflexmm_amd/libdfx.so`` (hand-written HIP kernels for gfx950) and raises when it is
missing -- there is no CPU fallback.  ``Engine`` is a thin, NumPy-only wrapper of one ``dfx_handle``.
"""
import ctypes as C
import os

import numpy as np

DFX_MAX_FNS = 2
DFX_FN_PARAMS = 5

BOND_LINEARIZED, BOND_NONLINEAR, BOND_SIMPLE_SPRING, BOND_STRETCH_TORSION = 0, 1, 2, 3
CONTACT_NONE, CONTACT_ANGLE, CONTACT_DISTANCE = 0, 1, 2
TABLEAU = {"dopri5": 0, "rk4": 1}
FN_ZERO, FN_PULSE, FN_HARMONIC, FN_RAMP, FN_SECH2TANH, FN_CONSTANT, FN_RAMP_CAP, FN_TABLE = range(8)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class dfx_special(C.Structure):
    _fields_ = [("block", C.c_int32), ("con_mask", C.c_int32),
                ("con_coef", (C.c_double * DFX_MAX_FNS) * 3), ("load_coef", (C.c_double * DFX_MAX_FNS) * 3)]


class dfx_problem(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_npb", C.c_int32), ("n_bonds", C.c_int32), ("bonds", _ip),
                ("bond_model", C.c_int32), ("contact", C.c_int32), ("n_special", C.c_int32),
                ("special", C.POINTER(dfx_special)), ("n_fns", C.c_int32), ("fn_type", C.c_int32 * DFX_MAX_FNS),
                ("batch", C.c_int32), ("tableau", C.c_int32), ("device", C.c_int32),
                ("fn_table_n", C.c_int32 * DFX_MAX_FNS), ("fn_table", _dp * DFX_MAX_FNS), ("streams", C.c_int32)]


_PARAM_FIELDS = ["centroid_node_vectors", "reference_vector", "k_bond", "inertia", "damping", "void_angle0",
                 "contact", "fn_params"]


class dfx_params(C.Structure):
    _fields_ = [(n, _dp) for n in _PARAM_FIELDS + ["block_centroids"]]


class dfx_grads(C.Structure):
    _fields_ = [(n, _dp) for n in _PARAM_FIELDS + ["state0", "block_centroids"]]


class dfx_stats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("rhs_evals", C.c_int64), ("launches", C.c_int64),
                ("kernel_ms", C.c_double), ("stage_kernel_us", C.c_double), ("streams", C.c_int64),
                ("stage_checkpoint", C.c_int64), ("checkpoint_records", C.c_int64), ("tile_kernels", C.c_int64)]


class dfx_design_map(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_npb", C.c_int32), ("n_bonds", C.c_int32), ("n_design", C.c_int32),
                ("base", _dp), ("gather", _ip), ("ref_points", _dp), ("bonds", _ip)]


EXPORTS = ["dfx_create", "dfx_destroy", "dfx_last_error", "dfx_set_params", "dfx_reserve", "dfx_forward", "dfx_forward_grid", "dfx_forward_grid_members",
           "dfx_forward_adaptive", "dfx_forward_adaptive_keep", "dfx_adaptive_step_counts", "dfx_adaptive_step_times", "dfx_adjoint",
           "dfx_objective_kinetic", "dfx_adjoint_kinetic", "dfx_kinetic_value_and_grad", "dfx_response_data", "dfx_rhs", "dfx_rhs_vjp", "dfx_energy",
           "dfx_device_count", "dfx_version", "dfx_share_checkpoint", "dfx_abi_layout", "dfx_member_status", "dfx_set_failure_policy",
           "dfx_test_set_spin_limit", "dfx_design_forward", "dfx_design_vjp"]
# multi-GPU collective (RCCL inside libdfx) and device helpers: HIP library only
COMM_EXPORTS = ["dfx_comm_unique_id", "dfx_comm_init", "dfx_comm_destroy", "dfx_comm_rccl_version", "dfx_comm_rank", "dfx_comm_size",
                "dfx_gather_objectives", "dfx_reduce_grads", "dfx_comm_allreduce", "dfx_comm_barrier", "dfx_comm_last_error",
                "dfx_mem_info", "dfx_device_synchronize", "dfx_kinetic_value_and_grad_device", "dfx_download", "dfx_forward_kinetic_value_and_grad"]
EXPORTS = EXPORTS + COMM_EXPORTS


def abi_layout():
    """sizeof + field offsets of the five public structs as THIS module mirrors them, in the order dfx_abi_layout reports."""
    out = []
    for cls in (dfx_special, dfx_problem, dfx_params, dfx_grads, dfx_stats):
        out.append(C.sizeof(cls))
        out += [getattr(cls, name).offset for name, _ in cls._fields_]
    return out


def check_abi_layout(lib):
    """A library whose structs are laid out differently from the ctypes mirrors above would read garbage silently: refuse it."""
    want = abi_layout()
    buf = (C.c_int32 * len(want))()
    lib.dfx_abi_layout.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    lib.dfx_abi_layout.restype = C.c_int
    n = lib.dfx_abi_layout(buf, len(want))
    if n != len(want) or list(buf) != want:
        raise RuntimeError(f"difflexmm_amd: struct layout of the library ({n} entries: {list(buf)[:n]}) differs from the binding's ({want}); "
                           "include/dfx.h and difflexmm_amd/_binding.py are out of step")


def declare(lib):
    """Attach argument / result types to every entry point of include/dfx.h."""
    H = C.c_void_p
    check_abi_layout(lib)
    lib.dfx_create.argtypes = [C.POINTER(dfx_problem), C.POINTER(H)]
    lib.dfx_destroy.argtypes = [H]
    if hasattr(lib, "dfx_share_checkpoint"):
        lib.dfx_share_checkpoint.argtypes = [H, H]
    lib.dfx_last_error.argtypes = [H]
    lib.dfx_last_error.restype = C.c_char_p
    lib.dfx_set_params.argtypes = [H, C.POINTER(dfx_params)]
    lib.dfx_reserve.argtypes = [H, C.c_int64, C.c_int32, C.c_int32]
    lib.dfx_forward.argtypes = [H, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, _dp, C.POINTER(dfx_stats)]
    lib.dfx_forward_grid.argtypes = [H, _dp, _dp, C.c_int32, _ip, _dp, C.c_int32, _dp, C.POINTER(dfx_stats)]
    lib.dfx_forward_grid_members.argtypes = [H, _dp, _dp, C.c_int32, _ip, _dp, C.c_int32, _dp, C.POINTER(dfx_stats)]
    lib.dfx_adaptive_step_counts.argtypes = [H, _ip]
    lib.dfx_adaptive_step_times.argtypes = [H, C.c_int32, _dp, C.c_int64, C.POINTER(C.c_int64)]
    lib.dfx_forward_adaptive.argtypes = [H, _dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_int64, _dp, C.POINTER(dfx_stats)]
    lib.dfx_forward_adaptive_keep.argtypes = [H, _dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_int32, _dp, C.POINTER(dfx_stats)]
    lib.dfx_design_forward.argtypes = [C.POINTER(dfx_design_map), _dp, C.c_int32, C.c_double, _dp, _dp, _dp, _dp]
    lib.dfx_design_vjp.argtypes = [C.POINTER(dfx_design_map), _dp, C.c_int32, C.c_double, _dp, _dp, _dp, _dp, _dp]
    lib.dfx_member_status.argtypes = [H, _ip]
    lib.dfx_set_failure_policy.argtypes = [H, C.c_int32]
    lib.dfx_test_set_spin_limit.argtypes = [H, C.c_int32]
    lib.dfx_adjoint.argtypes = [H, _dp, C.POINTER(dfx_grads), C.POINTER(dfx_stats)]
    lib.dfx_objective_kinetic.argtypes = [H, _ip, C.c_int32, _dp]
    lib.dfx_adjoint_kinetic.argtypes = [H, _ip, C.c_int32, C.POINTER(dfx_grads), C.POINTER(dfx_stats)]
    lib.dfx_kinetic_value_and_grad.argtypes = [H, _ip, C.c_int32, _dp, C.POINTER(dfx_grads), C.POINTER(dfx_grads), C.POINTER(dfx_stats)]
    lib.dfx_response_data.argtypes = [H, _dp, _dp, _dp, _dp]
    lib.dfx_rhs.argtypes = [H, _dp, C.c_double, _dp]
    lib.dfx_rhs_vjp.argtypes = [H, _dp, C.c_double, _dp, _dp, C.POINTER(dfx_grads)]
    lib.dfx_energy.argtypes = [H, _dp, _dp]
    lib.dfx_device_count.restype = C.c_int
    lib.dfx_version.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("dfx_last_error", "dfx_version", "dfx_comm_last_error") and (name not in COMM_EXPORTS or hasattr(lib, name)):
            getattr(lib, name).restype = C.c_int
    if hasattr(lib, "dfx_comm_init"):
        lib.dfx_comm_unique_id.argtypes = [C.c_char_p]
        lib.dfx_comm_init.argtypes = [C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(H)]
        for n in ("dfx_comm_destroy", "dfx_comm_rank", "dfx_comm_size", "dfx_comm_barrier"):
            getattr(lib, n).argtypes = [H]
        lib.dfx_comm_rccl_version.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.dfx_gather_objectives.argtypes = [H, _dp, C.c_int32, _dp]
        lib.dfx_reduce_grads.argtypes = [H, _dp, C.c_int64]
        lib.dfx_comm_allreduce.argtypes = [H, _dp, C.c_int64, C.c_int32]
        lib.dfx_comm_last_error.argtypes = []
        lib.dfx_comm_last_error.restype = C.c_char_p
        lib.dfx_mem_info.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.dfx_device_synchronize.argtypes = [C.c_int32]
        lib.dfx_kinetic_value_and_grad_device.argtypes = lib.dfx_kinetic_value_and_grad.argtypes
        lib.dfx_download.argtypes = [H, _dp, C.c_void_p, C.c_int64]
        lib.dfx_forward_kinetic_value_and_grad.argtypes = [H, _dp, _dp, C.c_int32, _ip, _ip, C.c_int32, _dp, C.POINTER(dfx_grads), C.POINTER(dfx_grads),
                                                           C.c_int32, C.POINTER(dfx_stats), C.POINTER(dfx_stats)]
    return lib


def mem_info(device=0, lib=None):
    """(free, total) bytes of HBM on a device."""
    lib = lib if lib is not None else load_library()
    f, t = C.c_int64(0), C.c_int64(0)
    if lib.dfx_mem_info(int(device), C.byref(f), C.byref(t)) != 0:
        raise RuntimeError("dfx_mem_info failed: " + lib.dfx_comm_last_error().decode())
    return f.value, t.value


def device_synchronize(device=0, lib=None):
    lib = lib if lib is not None else load_library()
    if lib.dfx_device_synchronize(int(device)) != 0:
        raise RuntimeError("dfx_device_synchronize failed: " + lib.dfx_comm_last_error().decode())


_LIB = None


def library_path():
    """``DFX_LIBRARY`` (a path) selects another build of the same HIP engine: kernel experiments side by side on one box."""
    return os.environ.get("DFX_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdfx.so")


def is_hip_engine(lib):
    """A build of the gfx950 engine says so in ``dfx_version()``; anything else with the same symbols (the CPU port of the oracle) is
    test infrastructure and must never stand in for the product."""
    try:
        lib.dfx_version.restype = C.c_char_p
        return b"gfx950" in lib.dfx_version()
    except Exception:       # noqa: BLE001
        return False


def check_engine_library(lib):
    """The product accepts only the HIP engine.  Test infrastructure hands over the CPU port by marking the handle it loaded
    (``oracle.cpu.load()`` sets ``lib._dfx_test_only = True``): an explicit, test-only door -- no environment variable opens it."""
    if is_hip_engine(lib) or getattr(lib, "_dfx_test_only", False) is True:
        return lib
    raise RuntimeError("difflexmm_amd: the library handed in is not the gfx950 engine (dfx_version() = %r): there is no CPU fallback"
                       % (getattr(lib, "dfx_version", lambda: b"?")(),))


def load_library():
    """Load the HIP engine.  Fails loudly when it has not been built, and when ``DFX_LIBRARY`` points at something that is not a
    build of the gfx950 engine: there is no fallback."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"difflexmm_amd: {path} not found -- build the HIP engine first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C difflexmm_amd/csrc)")
        lib = declare(C.CDLL(path))
        if not is_hip_engine(lib):
            raise RuntimeError(f"difflexmm_amd: {path} is not a build of the gfx950 engine (dfx_version() = {lib.dfx_version()!r}); "
                               "DFX_LIBRARY selects between builds of the HIP engine only")
        _LIB = lib
    return _LIB


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class Engine:
    """One ``dfx_handle``: a lattice + boundary-condition pattern, ``batch`` members wide."""

    def __init__(self, n_blocks, n_npb, bonds, bond_model, contact, special, fn_types, batch=1,
                 tableau="dopri5", device=0, lib=None, fn_tables=None, streams=0):
        self.lib = check_engine_library(lib) if lib is not None else load_library()
        self.n_blocks, self.n_npb, self.batch = int(n_blocks), int(n_npb), int(batch)
        self.bonds = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 2)
        self.n_bonds = len(self.bonds)
        self.fn_types = list(fn_types)
        self.n_fns = len(self.fn_types)
        self.contact = int(contact)
        spec = (dfx_special * max(1, len(special)))()
        for i, (block, mask, con, load) in enumerate(special):
            spec[i].block, spec[i].con_mask = int(block), int(mask)
            for d in range(3):
                for f in range(DFX_MAX_FNS):
                    spec[i].con_coef[d][f] = float(con[d][f]) if f < self.n_fns else 0.0
                    spec[i].load_coef[d][f] = float(load[d][f]) if f < self.n_fns else 0.0
        prob = dfx_problem()
        prob.n_blocks, prob.n_npb, prob.n_bonds = self.n_blocks, self.n_npb, self.n_bonds
        prob.bonds = self.bonds.ctypes.data_as(_ip)
        prob.bond_model, prob.contact = int(bond_model), self.contact
        prob.n_special, prob.special = len(special), spec
        prob.n_fns = self.n_fns
        for f in range(DFX_MAX_FNS):
            prob.fn_type[f] = int(self.fn_types[f]) if f < self.n_fns else 0
        prob.batch, prob.tableau, prob.device = self.batch, TABLEAU[tableau], int(device)
        prob.streams = int(streams or 0)
        tables = []                                   # (times, values) of FN_TABLE functions: static data, copied by dfx_create
        for f in range(DFX_MAX_FNS):
            tab = fn_tables[f] if fn_tables is not None and f < len(fn_tables) else None
            if tab is not None:
                tv = np.ascontiguousarray(np.concatenate([np.asarray(tab[0], dtype=float), np.asarray(tab[1], dtype=float)]))
                tables.append(tv)
                prob.fn_table_n[f], prob.fn_table[f] = len(tv) // 2, tv.ctypes.data_as(_dp)
            else:
                prob.fn_table_n[f], prob.fn_table[f] = 0, None
        self._h = C.c_void_p()
        rc = self.lib.dfx_create(C.byref(prob), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("dfx_create failed: " + self.lib.dfx_last_error(None).decode())
        self.n_timepoints = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dfx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: " + self.lib.dfx_last_error(self._h).decode())

    # -- parameters ---------------------------------------------------------------------------
    def shapes(self):
        B, nb, npb, nbd = self.batch, self.n_blocks, self.n_npb, self.n_bonds
        return {"centroid_node_vectors": (B, nb, npb, 2), "reference_vector": (B, nbd, 2), "k_bond": (B, nbd, 3),
                "inertia": (B, nb, 3), "damping": (B, nb, 3), "void_angle0": (B, nbd, 2), "contact": (B, 3),
                "fn_params": (B, max(1, self.n_fns), DFX_FN_PARAMS), "state0": (B, 2, nb, 3), "block_centroids": (B, nb, 2)}

    def set_params(self, **arrays):
        """Arrays by ``dfx_params`` field name; shapes as in :meth:`shapes` (broadcast over batch)."""
        sh = self.shapes()
        p = dfx_params()
        keep = []
        for name in _PARAM_FIELDS + ["block_centroids"]:
            a = arrays.get(name)
            if a is None:
                continue
            a = _f64(a, sh[name])
            keep.append(a)
            setattr(p, name, _ptr(a))
        self._check(self.lib.dfx_set_params(self._h, C.byref(p)), "dfx_set_params")

    def share_checkpoint(self, other):
        """Keep this engine's trajectory checkpoint in ``other``'s buffers (engines whose solves never overlap in time)."""
        self._check(self.lib.dfx_share_checkpoint(self._h, other._h), "dfx_share_checkpoint")

    def reserve(self, max_steps, max_timepoints, keep_trajectory=True):
        self._check(self.lib.dfx_reserve(self._h, int(max_steps), int(max_timepoints), int(bool(keep_trajectory))), "dfx_reserve")

    # -- solves -------------------------------------------------------------------------------
    def forward(self, state0, timepoints, steps_per_interval, keep_trajectory=False, want_fields=True, step_times=None):
        B, nb = self.batch, self.n_blocks
        state0 = _f64(state0, (B, 2, nb, 3)) if state0 is not None else None      # None: at rest (no upload)
        ts = _f64(timepoints)
        T = ts.shape[-1]
        fields = np.empty((B, T, 2, nb, 3)) if want_fields else None
        st = dfx_stats()
        if ts.ndim == 2:        # (batch, T): every member its own output times and step boundaries, shared step counts
            if ts.shape[0] != B:
                raise ValueError(f"per-member timepoints must be (batch={B}, T)")
            spis = np.ascontiguousarray(np.broadcast_to(steps_per_interval, (max(T - 1, 0),)), dtype=np.int32)
            n = int(spis.sum())
            if step_times is None:      # equal steps inside every member's own intervals
                step_times = np.stack([np.concatenate([a + (b - a) * np.arange(k) / k for a, b, k in zip(row[:-1], row[1:], spis)] + [row[-1:]])
                                       for row in ts])
            step_times = _f64(step_times, (B, n + 1))
            self._check(self.lib.dfx_forward_grid_members(self._h, _ptr(state0), _ptr(ts), T, spis.ctypes.data_as(_ip), _ptr(step_times),
                                                          int(bool(keep_trajectory)), _ptr(fields), C.byref(st)), "dfx_forward_grid_members")
        elif np.ndim(steps_per_interval) == 0 and step_times is None:
            self._check(self.lib.dfx_forward(self._h, _ptr(state0), _ptr(ts), T, int(steps_per_interval),
                                             int(bool(keep_trajectory)), _ptr(fields), C.byref(st)), "dfx_forward")
        else:   # its own number of equal steps in every output interval
            spis = np.ascontiguousarray(np.broadcast_to(steps_per_interval, (max(T - 1, 0),)), dtype=np.int32)
            if step_times is not None:      # caller-chosen step boundaries (every timepoint must be one of them)
                step_times = _f64(step_times, (int(spis.sum()) + 1,))
            self._check(self.lib.dfx_forward_grid(self._h, _ptr(state0), _ptr(ts), T, spis.ctypes.data_as(_ip), _ptr(step_times),
                                                  int(bool(keep_trajectory)), _ptr(fields), C.byref(st)), "dfx_forward_grid")
        self.n_timepoints = T
        return fields, _stats(st)

    def adaptive_step_counts(self):
        """(batch, T-1) accepted steps of the last forward_adaptive per member and output interval."""
        out = np.zeros((self.batch, max(self.n_timepoints - 1, 0)), dtype=np.int32)
        self._check(self.lib.dfx_adaptive_step_counts(self._h, out.ctypes.data_as(_ip)), "dfx_adaptive_step_counts")
        return out

    def adaptive_step_times(self, member=0):
        """End times of the steps one member accepted in the last forward_adaptive."""
        n = C.c_int64(0)
        self._check(self.lib.dfx_adaptive_step_times(self._h, int(member), None, 0, C.byref(n)), "dfx_adaptive_step_times")
        out = np.empty(n.value)
        self._check(self.lib.dfx_adaptive_step_times(self._h, int(member), _ptr(out), n.value, C.byref(n)), "dfx_adaptive_step_times")
        return out

    def forward_adaptive(self, state0, timepoints, rtol, atol, max_attempts=10_000_000, keep_trajectory=False, want_fields=True):
        """Adaptive Dormand-Prince with the reference's odeint semantics.  ``keep_trajectory``: the accepted steps are kept for the
        reverse sweep (``adjoint`` / ``kinetic_value_and_grad`` afterwards: the dense-output discrete adjoint of this very solve)."""
        B, nb = self.batch, self.n_blocks
        state0 = _f64(state0, (B, 2, nb, 3)) if state0 is not None else None        # None: every member starts at rest
        ts = _f64(timepoints)
        T = len(ts)
        fields = np.empty((B, T, 2, nb, 3)) if want_fields else None
        st = dfx_stats()
        if keep_trajectory or not want_fields:
            if not self.can_keep_adaptive:
                raise RuntimeError("this library has no dfx_forward_adaptive_keep")
            self._check(self.lib.dfx_forward_adaptive_keep(self._h, _ptr(state0), _ptr(ts), T, float(rtol), float(atol), int(max_attempts),
                                                           1 if keep_trajectory else 0, _ptr(fields), C.byref(st)), "dfx_forward_adaptive_keep")
        else:
            self._check(self.lib.dfx_forward_adaptive(self._h, _ptr(state0), _ptr(ts), T, float(rtol), float(atol),
                                                      int(max_attempts), _ptr(fields), C.byref(st)), "dfx_forward_adaptive")
        self.n_timepoints = T
        return fields, _stats(st)

    def member_status(self):
        """(batch,) int32: 0 ok, 1 non-finite, 2 step size underflow, 3 step budget exceeded -- of the last forward pass."""
        st = np.zeros(self.batch, dtype=np.int32)
        self._check(self.lib.dfx_member_status(self._h, st.ctypes.data_as(_ip)), "dfx_member_status")
        return st

    def set_failure_policy(self, isolate):
        """isolate=True: a member that diverges is flagged (``member_status``; its outputs NaN) instead of failing the call for the whole
        ensemble -- what the reference's list of forward problems does (problems/quads_focusing_multi_input.py:66-77)."""
        self._check(self.lib.dfx_set_failure_policy(self._h, 1 if isolate else 0), "dfx_set_failure_policy")

    @property
    def can_keep_adaptive(self):
        return hasattr(self.lib, "dfx_forward_adaptive_keep")

    def _grads(self, which):
        sh = self.shapes()
        g = dfx_grads()
        out = {}
        for name in which:
            if name == "fn_params" and self.n_fns == 0:
                continue
            if (name == "contact" and not self.contact) or (name == "void_angle0" and self.contact != CONTACT_ANGLE) or \
                    (name == "block_centroids" and self.contact != CONTACT_DISTANCE):
                continue
            out[name] = np.zeros(sh[name])
            setattr(g, name, _ptr(out[name]))
        return g, out

    ALL_GRADS = tuple(_PARAM_FIELDS + ["state0", "block_centroids"])

    def adjoint(self, fields_bar, which=ALL_GRADS):
        B, nb = self.batch, self.n_blocks
        if self.n_timepoints is None:
            raise RuntimeError("dfx_adjoint failed: run forward with keep_trajectory=1 first")
        fb = _f64(fields_bar, (B, self.n_timepoints, 2, nb, 3))
        g, out = self._grads(which)
        st = dfx_stats()
        self._check(self.lib.dfx_adjoint(self._h, _ptr(fb), C.byref(g), C.byref(st)), "dfx_adjoint")
        return out, _stats(st)

    def objective_kinetic(self, target_blocks):
        tb = np.ascontiguousarray(target_blocks, dtype=np.int32)
        obj = np.zeros(self.batch)
        self._check(self.lib.dfx_objective_kinetic(self._h, tb.ctypes.data_as(_ip), len(tb), _ptr(obj)),
                    "dfx_objective_kinetic")
        return obj

    def adjoint_kinetic(self, target_blocks, which=ALL_GRADS):
        tb = np.ascontiguousarray(target_blocks, dtype=np.int32)
        g, out = self._grads(which)
        st = dfx_stats()
        self._check(self.lib.dfx_adjoint_kinetic(self._h, tb.ctypes.data_as(_ip), len(tb), C.byref(g), C.byref(st)),
                    "dfx_adjoint_kinetic")
        return out, _stats(st)

    def kinetic_value_and_grad(self, target_blocks, which=ALL_GRADS, device=False):
        """objective (batch,) and the requested gradients in ONE call; the arrays are read-only views of library-owned pinned
        memory (valid until the next call on this engine: copy what must outlive it).  ``device=True`` leaves the gradients in HBM
        (HIP library only): the values are ``DeviceArray`` handles, ``.to_host()`` downloads one."""
        tb = np.ascontiguousarray(target_blocks, dtype=np.int32)
        sh = self.shapes()
        want, views = dfx_grads(), dfx_grads()
        names = [n for n in which if not (n == "fn_params" and self.n_fns == 0) and not (n == "contact" and not self.contact)
                 and not (n == "void_angle0" and self.contact != CONTACT_ANGLE)
                 and not (n == "block_centroids" and self.contact != CONTACT_DISTANCE)]
        flag = np.zeros(1)
        for n in names:
            setattr(want, n, _ptr(flag))
        obj = np.zeros(self.batch)
        st = dfx_stats()
        if device:
            if not hasattr(self.lib, "dfx_kinetic_value_and_grad_device"):
                raise RuntimeError("device-resident gradients need the HIP library")
            self._check(self.lib.dfx_kinetic_value_and_grad_device(self._h, tb.ctypes.data_as(_ip), len(tb), _ptr(obj), C.byref(want),
                                                                   C.byref(views), C.byref(st)), "dfx_kinetic_value_and_grad_device")
            return obj, {n: DeviceArray(self, C.cast(getattr(views, n), C.c_void_p).value, sh[n]) for n in names}, _stats(st)
        self._check(self.lib.dfx_kinetic_value_and_grad(self._h, tb.ctypes.data_as(_ip), len(tb), _ptr(obj), C.byref(want),
                                                        C.byref(views), C.byref(st)), "dfx_kinetic_value_and_grad")
        out = {}
        for n in names:
            a = np.ctypeslib.as_array(getattr(views, n), shape=sh[n])
            a.flags.writeable = False
            out[n] = a
        return obj, out, _stats(st)

    def forward_kinetic_value_and_grad(self, state0, timepoints, steps_per_interval, target_blocks, which=ALL_GRADS, device=False):
        """``forward(keep_trajectory=True, want_fields=False)`` + ``kinetic_value_and_grad`` as ONE library call (HIP library only): the host
        does not wait for the forward pass before it enqueues the reverse sweep.  Returns objective, gradients, forward stats, adjoint stats."""
        if not hasattr(self.lib, "dfx_forward_kinetic_value_and_grad"):
            raise RuntimeError("the fused call needs the HIP library")
        B, nb = self.batch, self.n_blocks
        state0 = _f64(state0, (B, 2, nb, 3)) if state0 is not None else None
        ts = _f64(timepoints)
        if ts.ndim != 1:
            raise ValueError("forward_kinetic_value_and_grad: one time grid for all members")
        T = len(ts)
        spis = np.ascontiguousarray(np.broadcast_to(steps_per_interval, (max(T - 1, 0),)), dtype=np.int32)
        tb = np.ascontiguousarray(target_blocks, dtype=np.int32)
        sh = self.shapes()
        want, views = dfx_grads(), dfx_grads()
        names = [n for n in which if not (n == "fn_params" and self.n_fns == 0) and not (n == "contact" and not self.contact)
                 and not (n == "void_angle0" and self.contact != CONTACT_ANGLE)
                 and not (n == "block_centroids" and self.contact != CONTACT_DISTANCE)]
        flag = np.zeros(1)
        for n in names:
            setattr(want, n, _ptr(flag))
        obj = np.zeros(B)
        st_f, st_a = dfx_stats(), dfx_stats()
        self._check(self.lib.dfx_forward_kinetic_value_and_grad(self._h, _ptr(state0), _ptr(ts), T, spis.ctypes.data_as(_ip), tb.ctypes.data_as(_ip),
                                                                len(tb), _ptr(obj), C.byref(want), C.byref(views), int(bool(device)),
                                                                C.byref(st_f), C.byref(st_a)), "dfx_forward_kinetic_value_and_grad")
        self.n_timepoints = T
        if device:
            out = {n: DeviceArray(self, C.cast(getattr(views, n), C.c_void_p).value, sh[n]) for n in names}
        else:
            out = {}
            for n in names:
                a = np.ctypeslib.as_array(getattr(views, n), shape=sh[n])
                a.flags.writeable = False
                out[n] = a
        return obj, out, _stats(st_f), _stats(st_a)

    def response_data(self, strains=True, kinetic=True):
        """Per-ligament strain energies (batch, T, n_bonds) x 3 and per-block kinetic energy (batch, T, n_blocks) of the last
        forward solve, computed on the device from its resident history."""
        B, T = self.batch, self.n_timepoints
        out = {}
        if strains:
            for n in ("strain_energy_stretch", "strain_energy_shear", "strain_energy_bending"):
                out[n] = np.empty((B, T, self.n_bonds))
        if kinetic:
            out["kinetic_energy"] = np.empty((B, T, self.n_blocks))
        self._check(self.lib.dfx_response_data(self._h, _ptr(out.get("strain_energy_stretch")), _ptr(out.get("strain_energy_shear")),
                                               _ptr(out.get("strain_energy_bending")), _ptr(out.get("kinetic_energy"))), "dfx_response_data")
        return out

    # -- test hooks ---------------------------------------------------------------------------
    def rhs(self, y, t):
        B, nb = self.batch, self.n_blocks
        y = _f64(y, (B, 2, nb, 3))
        dy = np.empty_like(y)
        self._check(self.lib.dfx_rhs(self._h, _ptr(y), float(t), _ptr(dy)), "dfx_rhs")
        return dy

    def rhs_vjp(self, y, t, lam, which=tuple(_PARAM_FIELDS + ["block_centroids"])):
        B, nb = self.batch, self.n_blocks
        y, lam = _f64(y, (B, 2, nb, 3)), _f64(lam, (B, 2, nb, 3))
        y_bar = np.empty_like(y)
        g, out = self._grads(which)
        self._check(self.lib.dfx_rhs_vjp(self._h, _ptr(y), float(t), _ptr(lam), _ptr(y_bar), C.byref(g)), "dfx_rhs_vjp")
        return y_bar, out

    def energy(self, u):
        u = _f64(u, (self.batch, self.n_blocks, 3))
        e = np.zeros(self.batch)
        self._check(self.lib.dfx_energy(self._h, _ptr(u), _ptr(e)), "dfx_energy")
        return e


class DeviceArray:
    """A float64 array in the HBM of an engine's GPU (a pointer the library handed out, valid until the next call on that engine)."""

    def __init__(self, engine, ptr, shape):
        self.engine, self.ptr, self.shape = engine, ptr, tuple(shape)
        self.size = int(np.prod(self.shape))
        self.nbytes = 8 * self.size

    def to_host(self):
        out = np.empty(self.shape)
        self.engine._check(self.engine.lib.dfx_download(self.engine._h, _ptr(out), self.ptr, self.size), "dfx_download")
        return out


def _stats(st):
    return {"steps": st.steps, "rhs_evals": st.rhs_evals, "launches": st.launches, "kernel_ms": st.kernel_ms,
            "stage_kernel_us": st.stage_kernel_us, "streams": st.streams, "stage_checkpoint": st.stage_checkpoint,
            "checkpoint_records": st.checkpoint_records, "tile_kernels": st.tile_kernels}
