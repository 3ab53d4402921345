"""Sharding of independent (design, input) solves over ranks and the single collective of the path.

Reference: the only multi-device construct of DifFlexMM is one ``pmap`` over independent forward inputs
(``problems/quads_kinetic_energy_static_tuning.py:454-478``) and the sequential list of forward problems in
``problems/quads_focusing_multi_input.py:66-77``.  Here: one process per GPU, members are dealt to ranks in contiguous equal
chunks, every rank integrates its members with no data-path communication, and the objectives (8 B per member) are
combined with ONE all-gather -- latency-bound, so ring vs tree is irrelevant.  Gradients of different designs stay on their
rank; gradients w.r.t. a SHARED design (multi-input problems) are summed with one all-reduce.

The collective is native: ``RcclComm`` calls ``dfx_comm_init / dfx_gather_objectives / dfx_reduce_grads`` of libdfx (RCCL
over xGMI inside the library, ``include/dfx.h``); the unique id travels from rank 0 to the others over a TCP control channel, over
which the ranks also agree on go / no-go before any of them enters ``ncclCommInitRank``.  ``SocketComm`` is a plain-TCP stand-in
with the same methods for rehearsing the N > 1 path where RCCL cannot run (several ranks on ONE GPU, or no GPU at all);
any object with ``rank``, ``world``, ``all_gather``, ``all_reduce`` and ``barrier`` can be passed as ``comm=``.
No PyTorch anywhere in this module.
"""
import ctypes as C
import os
import pickle
import socket
import struct
import time

import numpy as np

_DEFAULT = None          # communicator used when a function is called without comm=


class SerialComm:
    rank, world = 0, 1

    def all_gather(self, local):
        return np.asarray(local, dtype=np.float64)[None].copy()

    def all_reduce(self, array, op="sum"):
        return np.array(array, dtype=np.float64, copy=True)

    def barrier(self):
        pass

    def close(self):
        pass


class RcclComm:
    """RCCL over xGMI through libdfx (one rank per GPU).

    Bring-up is agreed on BEFORE any rank enters ``ncclCommInitRank`` (a rank that fails alone would leave the others blocked inside
    it): with a control channel ``ctrl`` (any communicator with ``all_gather`` / ``all_reduce``; ``init_from_env`` opens a
    ``SocketComm``) rank 0 sends the unique id over it, every rank sets its device, and all ranks exchange a go / no-go -- one failed
    preflight and every rank raises.  Ranks must sit on distinct devices.  Without a control channel the id travels through
    ``uid_file`` (rank 0 removes a stale file first and publishes atomically)."""

    def __init__(self, rank, world, device, uid_file=None, lib=None, timeout=300.0, ctrl=None):
        from ._binding import load_library
        self.lib = lib if lib is not None else load_library()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        uid = C.create_string_buffer(128)
        err = ""
        if ctrl is not None:
            if self.rank == 0 and self.lib.dfx_comm_unique_id(uid) != 0:
                err = "dfx_comm_unique_id: " + self.lib.dfx_comm_last_error().decode()
            sent = ctrl.all_gather(np.frombuffer(uid.raw, dtype=np.uint8).astype(np.float64))     # rank 0's row is the id
            uid = C.create_string_buffer(bytes(np.asarray(sent[0], dtype=np.uint8)), 128)
            f, t = C.c_int64(0), C.c_int64(0)
            if not err and self.lib.dfx_mem_info(self.device, C.byref(f), C.byref(t)) != 0:       # sets the device: the preflight
                err = f"device {self.device}: " + self.lib.dfx_comm_last_error().decode()
            devices = ctrl.all_gather(np.array([float(self.device)])).ravel()
            if not err and len(set(devices.tolist())) != self.world and os.environ.get("DFX_ALLOW_SHARED_DEVICE") != "1":
                err = f"ranks share a device ({devices.astype(int).tolist()}): RCCL needs one GPU per rank"
            ok = ctrl.all_reduce([0.0 if err else 1.0], "min")[0] > 0
            if not ok:
                raise RuntimeError("RcclComm: bring-up refused before ncclCommInitRank (" + (err or "preflight failed on another rank") + ")")
        elif self.rank == 0:
            try:
                os.remove(uid_file)                      # a crashed launch with the same name may have left one behind
            except OSError:
                pass
            self._check(self.lib.dfx_comm_unique_id(uid), "dfx_comm_unique_id")
            tmp = f"{uid_file}.tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(uid.raw)
            os.replace(tmp, uid_file)                     # atomic: readers see nothing or all 128 bytes
        else:
            t0 = time.time()
            while not (os.path.exists(uid_file) and os.path.getsize(uid_file) == 128):
                if time.time() - t0 > timeout:
                    raise RuntimeError(f"RcclComm: rank 0 did not publish {uid_file} within {timeout} s")
                time.sleep(0.01)
            with open(uid_file, "rb") as f:
                uid = C.create_string_buffer(f.read(), 128)
        self._c = C.c_void_p()
        self._check(self.lib.dfx_comm_init(self.rank, self.world, uid, self.device, C.byref(self._c)), "dfx_comm_init")
        self.barrier()
        if ctrl is None and self.rank == 0:
            try:
                os.remove(uid_file)
            except OSError:
                pass

    def info(self):
        """What the collective really is: ranks RCCL sees, RCCL version running / compiled against."""
        rt, cp = C.c_int32(0), C.c_int32(0)
        self.lib.dfx_comm_rccl_version(C.byref(rt), C.byref(cp))
        return {"ranks_seen": int(self.lib.dfx_comm_size(self._c)), "rccl_runtime": int(rt.value), "rccl_compiled": int(cp.value)}

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: " + self.lib.dfx_comm_last_error().decode())

    def all_gather(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64).ravel()
        out = np.empty((self.world, local.size))
        dp = C.POINTER(C.c_double)
        self._check(self.lib.dfx_gather_objectives(self._c, local.ctypes.data_as(dp), local.size, out.ctypes.data_as(dp)),
                    "dfx_gather_objectives")
        return out

    def all_reduce(self, array, op="sum"):
        a = np.array(array, dtype=np.float64, copy=True, order="C")
        dp = C.POINTER(C.c_double)
        if op == "sum":
            self._check(self.lib.dfx_reduce_grads(self._c, a.ctypes.data_as(dp), a.size), "dfx_reduce_grads")
        else:
            self._check(self.lib.dfx_comm_allreduce(self._c, a.ctypes.data_as(dp), a.size, {"max": 1, "min": 2}[op]),
                        "dfx_comm_allreduce")
        return a

    def barrier(self):
        self._check(self.lib.dfx_comm_barrier(self._c), "dfx_comm_barrier")

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self.lib.dfx_comm_destroy(self._c)
            self._c = C.c_void_p()


class SocketComm:
    """The same three operations over TCP through rank 0 (a star; payloads are a few bytes to a few KB).  A stand-in for
    rehearsals and CPU tests -- not the production collective."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29511, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = []
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[self._recv(conn)] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=timeout)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._send(s, self.rank)
            self._peers = [s]

    @staticmethod
    def _send(sock, obj):
        data = pickle.dumps(obj, protocol=4)
        sock.sendall(struct.pack("<Q", len(data)) + data)

    @staticmethod
    def _recv(sock):
        def read(n):
            buf = b""
            while len(buf) < n:
                chunk = sock.recv(n - len(buf))
                if not chunk:
                    raise ConnectionError("SocketComm: peer closed the connection")
                buf += chunk
            return buf
        (n,) = struct.unpack("<Q", read(8))
        return pickle.loads(read(n))

    def _exchange(self, local, combine):
        if self.world == 1:
            return combine([local])
        if self.rank == 0:
            parts = [local] + [self._recv(p) for p in self._peers]
            out = combine(parts)
            for p in self._peers:
                self._send(p, out)
            return out
        self._send(self._peers[0], local)
        return self._recv(self._peers[0])

    def all_gather(self, local):
        return self._exchange(np.asarray(local, dtype=np.float64).ravel(), lambda parts: np.stack(parts))

    def all_reduce(self, array, op="sum"):
        f = {"sum": np.sum, "max": np.max, "min": np.min}[op]
        return self._exchange(np.asarray(array, dtype=np.float64), lambda parts: f(np.stack(parts), axis=0))

    def barrier(self):
        self._exchange(0.0, lambda parts: 0.0)

    def close(self):
        for p in self._peers:
            try:
                p.close()
            except OSError:
                pass
        self._peers = []


def init_from_env(backend="rccl", device=None, lib=None, ctrl=None):
    """Communicator of this process from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT --
    what ``torch.distributed.run`` and ``bench.py``'s own launcher export) and make it the module default.  ``backend="rccl"``
    opens a TCP control channel first (or uses ``ctrl``): the RCCL bring-up is agreed on over it."""
    global _DEFAULT
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world_size == 1:
        _DEFAULT = SerialComm()
    elif backend == "rccl":
        own = ctrl is None
        if own:
            ctrl = SocketComm(rank, world_size, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                              int(os.environ.get("DFX_SOCKET_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 11)))
        try:
            _DEFAULT = RcclComm(rank, world_size, local_rank if device is None else device, lib=lib, ctrl=ctrl)
        finally:
            if own:
                ctrl.close()
    elif backend == "socket":
        _DEFAULT = SocketComm(rank, world_size, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                              int(os.environ.get("DFX_SOCKET_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 11)))
    else:
        raise ValueError(f"unknown backend {backend!r} (rccl | socket)")
    return _DEFAULT


def set_default(comm):
    global _DEFAULT
    _DEFAULT = comm


def _comm(comm):
    if comm is not None:
        return comm
    return _DEFAULT if _DEFAULT is not None else SerialComm()


def world(comm=None):
    c = _comm(comm)
    return c.rank, c.world


def shard_bounds(n_items, rank, world_size):
    """Contiguous chunk [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_objectives(local_values, n_total, comm=None):
    """All ranks receive the (n_total,) vector of objectives in member order (ONE all-gather of equal-size slots)."""
    c = _comm(comm)
    local_values = np.atleast_1d(np.asarray(local_values, dtype=np.float64))
    if c.world == 1:
        return local_values
    width = -(-n_total // c.world)
    buf = np.zeros(width)
    buf[:len(local_values)] = local_values
    out = c.all_gather(buf)
    parts = []
    for r in range(c.world):
        lo, hi = shard_bounds(n_total, r, c.world)
        parts.append(out[r][:hi - lo])
    return np.concatenate(parts)


def sum_shared_gradients(arrays, comm=None):
    """Sum gradient arrays of a design shared by all ranks (one all-reduce over the flattened concatenation)."""
    c = _comm(comm)
    if c.world == 1:
        return arrays
    flat = c.all_reduce(np.concatenate([np.ravel(a) for a in arrays]), "sum")
    out, pos = [], 0
    for a in arrays:
        out.append(flat[pos:pos + a.size].reshape(np.shape(a)))
        pos += a.size
    return out


def _engine_of(objective):
    fw = getattr(objective, "forward", None)
    sd = getattr(fw, "solve_dynamics", None)
    return getattr(sd, "engine", None)


def evaluate_ensemble(objective, designs, comm=None, with_status=False):
    """Every rank evaluates ``objective.value_and_grad`` on its chunk of ``designs`` (as one batch if the forward
    problem was set up with ``batch`` = chunk size, else one by one) and returns (all objectives, local gradients,
    (lo, hi)) -- with ``with_status=True`` also the status of every design, on every rank.

    Failure isolation (in the reference a diverging member of the list of forward problems,
    problems/quads_focusing_multi_input.py:66-77, yields NaN for that member only): the engine flags a member whose state
    becomes non-finite or whose step underflows instead of failing the call (``dfx_set_failure_policy``), its objective
    comes back as NaN with status 1 / 2 / 3 (``dfx_member_status``), the others are untouched.  A rank whose evaluation
    RAISES still enters the collective -- with NaN objectives and status -1 for its designs -- so no peer is left blocked in
    the all-gather; the exception is raised again after the collective unless ``with_status`` asked for the statuses."""
    c = _comm(comm)
    lo, hi = shard_bounds(len(designs), c.rank, c.world)
    mine = list(designs[lo:hi])
    batch = getattr(objective.forward, "batch", 1)
    vals, grads = [], []
    status = np.zeros(len(mine))
    eng = _engine_of(objective)
    if eng is not None and hasattr(eng, "set_failure_policy"):
        eng.set_failure_policy(True)
    err = None
    try:
        if batch > 1:
            assert len(mine) == batch, "chunk size must equal the solver's batch"
            v, g = objective.value_and_grad(mine)
            vals, grads = list(np.atleast_1d(v)), list(g)
            if eng is not None and hasattr(eng, "member_status"):
                status[:] = eng.member_status()
        else:
            for i, dsg in enumerate(mine):
                v, g = objective.value_and_grad(dsg)
                vals.append(float(v)); grads.append(g)
                if eng is not None and hasattr(eng, "member_status"):
                    status[i] = eng.member_status()[0]
    except Exception as e:          # noqa: BLE001 -- reported through the statuses (and raised again below), never swallowed
        err = e
        vals = list(vals) + [np.nan] * (len(mine) - len(vals))
        grads = list(grads) + [None] * (len(mine) - len(grads))
        status[:] = np.where(np.isnan(np.asarray(vals, dtype=float)), -1, status)
    vals = np.asarray(vals, dtype=float)
    vals = np.where(status != 0, np.nan, vals)            # a flagged member's objective is NaN whatever arithmetic produced
    all_vals = gather_objectives(vals, len(designs), c) if c.world == 1 else None
    if c.world == 1:
        all_status = status.astype(int)
    else:       # ONE all-gather carries (value, status) of every design
        width = -(-len(designs) // c.world)
        buf = np.zeros(2 * width)
        buf[:len(vals)] = vals
        buf[width:width + len(status)] = status
        out = c.all_gather(buf)
        pv, ps = [], []
        for r in range(c.world):
            rlo, rhi = shard_bounds(len(designs), r, c.world)
            pv.append(out[r][:rhi - rlo]); ps.append(out[r][width:width + rhi - rlo])
        all_vals, all_status = np.concatenate(pv), np.concatenate(ps).astype(int)
    if with_status:
        return all_vals, grads, (lo, hi), all_status
    if err is not None:
        raise err
    return all_vals, grads, (lo, hi)
