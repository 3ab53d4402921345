"""world_size-2 tests of the N>1 path on CPU: designs sharded over ranks, no data-path collective, objectives combined by one
all-gather, shared-design gradients by one all-reduce.  Two communicators drive the same `difflexmm_amd.ensemble` code:
a gloo adapter (torch.distributed lives only here, in the tests) and the package's plain-TCP `SocketComm`.  The production
communicator (`RcclComm`, RCCL inside libdfx) has the same three methods; it needs one GPU per rank."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(lib):
    import math
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    fw = QuadsFocusingForward(
        n1_blocks=6, n2_blocks=6, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5,
        density=6.18e-9, damping=1e-4 * np.ones((36, 3)), amplitude=7.5, loading_rate=3000.0, input_delay=1e-5,
        n_excited_blocks=2, loaded_side="left", input_shift=0, simulation_time=4e-4, n_timepoints=5,
        use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180, cutoff_angle=42 * math.pi / 180,
        steps_per_interval=10, _lib=lib)
    fw.setup()
    obj = TargetKineticEnergy(fw, (2, 2), (1, 1))
    designs = []
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
        designs.append(tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base))
    return fw, obj, designs


class GlooComm:
    """torch.distributed (gloo) behind the communicator interface of difflexmm_amd.ensemble."""

    def __init__(self, rank, world_size, port):
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        self.dist, self.rank, self.world = dist, rank, world_size

    def all_gather(self, local):
        import torch
        mine = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float64).ravel())
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return np.stack([o.numpy() for o in out])

    def all_reduce(self, array, op="sum"):
        import torch
        t = torch.as_tensor(np.array(array, dtype=np.float64))
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}[op])
        return t.numpy()

    def barrier(self):
        self.dist.barrier()

    def close(self):
        self.dist.destroy_process_group()


def _worker(rank, world_size, port, out_dir, kind):
    sys.path.insert(0, ROOT)
    from difflexmm_amd import ensemble
    from oracle.cpu import load
    comm = GlooComm(rank, world_size, port) if kind == "gloo" else ensemble.SocketComm(rank, world_size, "127.0.0.1", port)
    ensemble.set_default(comm)
    fw, obj, designs = _make(load())
    values, grads, (lo, hi) = ensemble.evaluate_ensemble(obj, designs)
    shared = ensemble.sum_shared_gradients([np.full((2, 3), float(rank + 1)), np.arange(4.0) * (rank + 1)])
    slowest = comm.all_reduce([float(rank)], "max")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), values=values, lo=lo, hi=hi, g0=grads[0][0], s0=shared[0], s1=shared[1],
             slowest=slowest)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("kind", ["gloo", "socket"])
def test_two_ranks_shard_designs_and_gather_objectives(tmp_path, cpu_lib, kind):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + (0 if kind == "gloo" else 2017)
    mp.spawn(_worker, args=(2, port, str(tmp_path), kind), nprocs=2, join=True)
    fw, obj, designs = _make(cpu_lib)
    ref = [obj.value_and_grad(d) for d in designs]
    ref_vals = np.array([r[0] for r in ref])
    assert np.all(ref_vals > 0)
    for rank in range(2):
        d = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_allclose(d["values"], ref_vals, rtol=1e-12)          # every rank holds all objectives
        assert (int(d["lo"]), int(d["hi"])) == ((0, 2), (2, 4))[rank]
        np.testing.assert_allclose(d["g0"], ref[int(d["lo"])][1][0], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(d["s0"], np.full((2, 3), 3.0))               # 1 + 2
        np.testing.assert_allclose(d["s1"], np.arange(4.0) * 3)
        assert float(d["slowest"][0]) == 1.0


def _collapsed(fw, design, blk=14, keep=1e-5):
    """`design` with block `blk` of the 6 x 6 lattice shrunk to (almost) a point: its mass is ~1e-10 of its neighbours', the explicit
    integrator blows up within a few steps -- a member that diverges for a physical reason, not a malformed input."""
    d = tuple(np.array(a) for a in design)
    cnv = fw.geometry.centroid_node_vectors(*d)[blk]
    f = 1 - keep
    d[0][3, 2] -= f * cnv[0]; d[1][2, 3] -= f * cnv[1]; d[0][2, 2] -= f * cnv[2]; d[1][2, 2] -= f * cnv[3]
    return d


def _isolation_worker(rank, world_size, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    from difflexmm_amd import ensemble
    from oracle.cpu import load
    comm = GlooComm(rank, world_size, port)
    ensemble.set_default(comm)
    fw, obj, designs = _make(load())
    designs = list(designs)
    if mode == "diverges":
        designs[3] = _collapsed(fw, designs[3])                      # member 1 of rank 1
    else:
        bad = tuple(np.array(a) for a in designs[2]); bad[0][1, 1, 0] = np.nan
        designs[2] = bad                                             # rank 1's first evaluation raises (set_params refuses the NaN)
    values, grads, (lo, hi), status = ensemble.evaluate_ensemble(obj, designs, with_status=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), values=values, status=status, lo=lo, hi=hi)
    raised = ""
    if mode == "raises":      # without with_status the exception comes back -- AFTER the collective, so the peer is not left blocked in it
        try:
            ensemble.evaluate_ensemble(obj, designs)
        except Exception as e:      # noqa: BLE001
            raised = type(e).__name__
    with open(os.path.join(out_dir, f"raised{rank}"), "w") as f:
        f.write(raised)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("mode", ["diverges", "raises"])
def test_a_failing_member_does_not_take_the_ensemble_with_it(tmp_path, cpu_lib, mode):
    """SURVEY section 5 / round-5 verdict item 4: in the reference a diverging member of the list of forward problems
    (problems/quads_focusing_multi_input.py:66-77) yields NaN for that member only.  world_size 2 over gloo: one member of rank 1
    diverges (a block with almost no mass: the state overflows) -- both ranks return, the member is NaN and flagged on BOTH ranks, the
    other three objectives equal the serial run.  And a rank whose evaluation raises still enters the all-gather (status -1 for its
    designs): the peer is not left waiting in the collective."""
    import torch.multiprocessing as mp
    port = 35000 + (os.getpid() % 2000) + (0 if mode == "diverges" else 7)
    mp.spawn(_isolation_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    fw, obj, designs = _make(cpu_lib)
    ref_vals = np.array([obj.value_and_grad(d)[0] for d in designs])
    for rank in range(2):
        d = np.load(tmp_path / f"rank{rank}.npz")
        if mode == "diverges":
            assert d["status"].tolist() == [0, 0, 0, 1]
            assert np.isnan(d["values"][3])
            np.testing.assert_allclose(d["values"][:3], ref_vals[:3], rtol=1e-12)
        else:
            assert d["status"].tolist() == [0, 0, -1, -1]                      # rank 1 raised at its first design: both of its are void
            assert np.all(np.isnan(d["values"][2:]))
            np.testing.assert_allclose(d["values"][:2], ref_vals[:2], rtol=1e-12)
            assert open(tmp_path / f"raised{rank}").read() == ("" if rank == 0 else "RuntimeError")


def test_shard_bounds_cover_everything():
    from difflexmm_amd.ensemble import shard_bounds
    for n in (1, 7, 64, 256):
        for ws in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, ws) for r in range(ws)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


class _FakeRccl:
    """Stands in for libdfx's comm entry points in the bring-up test below (no GPU here): records whether ncclCommInitRank would
    have been entered and with which unique id."""

    def __init__(self, rank, world, out_dir, fail_rank):
        self.rank, self.world, self.out_dir, self.fail_rank = rank, world, out_dir, fail_rank

    def dfx_comm_unique_id(self, buf):
        buf.raw = bytes((7 * i + 3) % 251 for i in range(128))
        return 0

    def dfx_mem_info(self, device, f, t):
        return 2 if self.rank == self.fail_rank else 0

    def dfx_comm_last_error(self):
        return b"no such HIP device (fake)"

    def dfx_comm_init(self, rank, world, uid, device, out):
        with open(os.path.join(self.out_dir, f"entered{rank}"), "wb") as f:
            f.write(uid.raw)
        return 0

    def dfx_comm_barrier(self, c):
        return 0

    def dfx_comm_size(self, c):
        return self.world

    def dfx_comm_rccl_version(self, rt, cp):
        rt._obj.value, cp._obj.value = 22606, 22707
        return 0

    def dfx_comm_destroy(self, c):
        return 0


def _bringup_worker(rank, world_size, port, out_dir, fail_rank, same_device):
    sys.path.insert(0, ROOT)
    from difflexmm_amd import ensemble
    ctrl = ensemble.SocketComm(rank, world_size, "127.0.0.1", port)
    res = "ok"
    try:
        comm = ensemble.RcclComm(rank, world_size, 0 if same_device else rank, lib=_FakeRccl(rank, world_size, out_dir, fail_rank), ctrl=ctrl)
        res = "ok " + repr(comm.info())
    except RuntimeError as e:
        res = "refused: " + str(e)
    with open(os.path.join(out_dir, f"result{rank}"), "w") as f:
        f.write(res)
    ctrl.barrier()
    ctrl.close()


@pytest.mark.parametrize("mode", ["ok", "one rank fails its preflight", "ranks share a device"])
def test_rccl_bringup_is_agreed_on_before_any_rank_enters_init(tmp_path, mode):
    """Round-2 advice: a rank that failed before ncclCommInitRank left the others blocked inside it.  Now the unique id travels
    over the control channel and all ranks exchange go / no-go first: one failed preflight (or two ranks on one device) and EVERY
    rank raises without entering the init call; otherwise all enter it with rank 0's id."""
    import torch.multiprocessing as mp
    port = 31000 + (os.getpid() % 2000) + {"ok": 0, "one rank fails its preflight": 1, "ranks share a device": 2}[mode]
    mp.spawn(_bringup_worker, args=(2, port, str(tmp_path), 1 if mode.startswith("one") else -1, mode.startswith("ranks")), nprocs=2, join=True)
    results = [open(tmp_path / f"result{r}").read() for r in range(2)]
    if mode == "ok":
        assert all(r.startswith("ok") and "'ranks_seen': 2" in r and "22606" in r for r in results), results
        ids = [open(tmp_path / f"entered{r}", "rb").read() for r in range(2)]
        assert ids[0] == ids[1] == bytes((7 * i + 3) % 251 for i in range(128))
    else:
        assert all(r.startswith("refused: RcclComm: bring-up refused before ncclCommInitRank") for r in results), results
        assert not any((tmp_path / f"entered{r}").exists() for r in range(2))
        assert ("no such HIP device" in results[1]) if mode.startswith("one") else ("share a device" in results[0])


def _bench_comm_worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    import argparse
    import json
    import bench
    comm, info = bench.make_comm(argparse.Namespace(backend="socket"), world_size, rank, rank)
    gathered = comm.all_gather([float(rank)])
    with open(os.path.join(out_dir, f"info{rank}.json"), "w") as f:
        json.dump(dict(info, gathered=gathered.ravel().tolist()), f)
    comm.barrier()
    comm.close()


def test_bench_reports_what_the_collective_really_is(tmp_path):
    """bench.py --gpus N puts the collective and the number of ranks it saw into the JSON line (`config.collective`, `ranks_seen`)."""
    import json
    import torch.multiprocessing as mp
    port = 33000 + (os.getpid() % 2000)
    mp.spawn(_bench_comm_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        info = json.load(open(tmp_path / f"info{r}.json"))
        assert info["collective"] == "socket (rehearsal)" and info["ranks_seen"] == 2 and info["gathered"] == [0.0, 1.0]
