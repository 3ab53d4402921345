"""Sharding of independent (design, input) solves over ranks and the single collective of the path.

Reference: the only multi-device construct of DifFlexMM is one ``pmap`` over independent forward inputs
(``problems/quads_kinetic_energy_static_tuning.py:454-478``) and the sequential list of forward problems in
``problems/quads_focusing_multi_input.py:66-77``.  Here: one process per GPU (``torch.distributed``; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests), members are dealt to ranks in contiguous equal chunks, every rank
integrates its members with no data-path communication, and the objectives (8 B per member) are combined with ONE
``all_gather`` -- latency-bound, so ring vs tree is irrelevant.  Gradients of different designs stay on their rank;
gradients w.r.t. a SHARED design (multi-input problems) are summed with one ``all_reduce``.
"""
import numpy as np


def _dist():
    import sys
    if "torch.distributed" not in sys.modules:      # nobody initialised a process group: single process, no torch import
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard_bounds(n_items, rank, world_size):
    """Contiguous chunk [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _tensor(x):
    import torch
    d = _dist()
    t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64))
    if d is not None and d.get_backend() == "nccl":
        t = t.cuda()
    return t


def gather_objectives(local_values, n_total):
    """All ranks receive the (n_total,) vector of objectives in member order."""
    import torch
    d = _dist()
    local_values = np.atleast_1d(np.asarray(local_values, dtype=np.float64))
    if d is None:
        return local_values
    rank, ws = d.get_rank(), d.get_world_size()
    width = -(-n_total // ws)                      # equal-size slots so one all_gather suffices
    buf = np.zeros(width)
    buf[:len(local_values)] = local_values
    mine = _tensor(buf)
    out = [torch.empty_like(mine) for _ in range(ws)]
    d.all_gather(out, mine)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(n_total, r, ws)
        parts.append(out[r].cpu().numpy()[:hi - lo])
    return np.concatenate(parts)


def sum_shared_gradients(arrays):
    """Sum gradient arrays of a design shared by all ranks (one all_reduce over the flattened concatenation)."""
    import torch
    d = _dist()
    if d is None:
        return arrays
    flat = _tensor(np.concatenate([np.ravel(a) for a in arrays]))
    d.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    flat = flat.cpu().numpy()
    out, pos = [], 0
    for a in arrays:
        out.append(flat[pos:pos + a.size].reshape(np.shape(a)))
        pos += a.size
    return out


def evaluate_ensemble(objective, designs):
    """Every rank evaluates ``objective.value_and_grad`` on its chunk of ``designs`` (as one batch if the forward
    problem was set up with ``batch`` = chunk size, else one by one) and returns (all objectives, local gradients,
    (lo, hi))."""
    rank, ws = world()
    lo, hi = shard_bounds(len(designs), rank, ws)
    mine = list(designs[lo:hi])
    batch = getattr(objective.forward, "batch", 1)
    vals, grads = [], []
    if batch > 1:
        assert len(mine) == batch, "chunk size must equal the solver's batch"
        v, g = objective.value_and_grad(mine)
        vals, grads = list(np.atleast_1d(v)), list(g)
    else:
        for dsg in mine:
            v, g = objective.value_and_grad(dsg)
            vals.append(float(v)); grads.append(g)
    return gather_objectives(vals, len(designs)), grads, (lo, hi)
