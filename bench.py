#!/usr/bin/env python3
"""bench.py -- timesteps x rigid-units / s of the hot path (forward + discrete adjoint) on MI355X.

Workload (BASELINE.json configs[2], SURVEY 8(d) "C3"): 128x128 quad lattice (16 384 rigid units), nonlinear
ligaments + viscous damping + angle-based contact, raised-cosine displacement pulse on 2 left-edge blocks,
clamped corners, Dormand-Prince tableau on a fixed grid with dt = (2/f)/50 000, one output every 250 steps,
objective = kinetic energy of the 2x2 target blocks, gradient w.r.t. the 66 048 geometry parameters.
One "step" = one RK step (6 RHS evaluations) of one member, forward AND reverse.  `--steps K` times exactly K
steps (output intervals of 250 steps and, if K is not a multiple of 250, one shorter last interval); the full config is
K = 50 000.  Default: K = 5 000 with 16 independent designs per GPU, the largest member count whose state checkpoint AND
stage checkpoint (72 + 120 B per unit and step: the reverse sweep then needs no recompute launches) fit the 288 GB;
longer runs fall back to the state checkpoint alone (K = 10 000) and then to fewer members (K = 50 000: 4).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--members M] [--size 128]

N > 1: launched by torch.distributed.run, one rank per GPU; every rank integrates its own design (weak
scaling, no data-path collective), objectives are gathered with one RCCL all_gather.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPI = 250                     # steps between outputs: 50 000 steps / 200 output intervals
FREQ = 30.0
DT = (2.0 / FREQ) / 50000.0
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md, chip-level parameters
# algorithmic bytes per rigid unit per launch
BYTES_FWD_STAGE = 272 + 72    # one RHS evaluation (quads + contact) + its share of the stage combine (432/6)
BYTES_ADJ_STAGE = 272 + 96 + 256  # stage data + lambda/Ybar read-write + parameter-gradient RMW
# with the stage checkpoint 5 of the 6 reverse launches of a step also rebuild the next stage record: the forward's stage
# combine (72 B per stage, SURVEY 8(d): 432 B per 6 stages) moves into the reverse launch instead of a recompute launch
BYTES_ADJ_STAGE_REBUILD = BYTES_ADJ_STAGE + 72 * 5 / 6


def c3_problem(size, seed, members, lib=None, device=0):
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    spacing, bond = 15.0, 2.25
    rho, ksh, kr = 6.18e-9, 1.19, 1.5
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh),
                                 2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh),
                                 2 * math.sqrt(0.02175026 * rho * spacing ** 4 * kr)]) * np.ones((size * size, 1))
    fw = QuadsFocusingForward(
        n1_blocks=size, n2_blocks=size, spacing=spacing, bond_length=bond, k_stretch=120.0, k_shear=ksh, k_rot=kr,
        density=rho, damping=damping, amplitude=7.5, loading_rate=FREQ, input_delay=0.1 / FREQ, n_excited_blocks=2,
        loaded_side="left", input_shift=0, simulation_time=2.0 / FREQ, n_timepoints=201, use_contact=True,
        k_contact=1.5, min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180, steps_per_interval=SPI,
        batch=members, device=device, _lib=lib)
    fw.setup()
    obj = TargetKineticEnergy(fw, (2, 2), (size // 6, size // 5))
    designs = []
    for m in range(members):
        rng = np.random.default_rng(seed + m)
        base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
        designs.append(tuple(b + rng.uniform(-0.02 * spacing, 0.02 * spacing, b.shape) for b in base))
    return fw, obj, designs


def step_grid(n_steps, spi=SPI):
    """EXACTLY n_steps RK steps of size DT: full output intervals of `spi` steps and, when n_steps is not a multiple of
    spi, one shorter last interval.  Returns (timepoints, steps per interval)."""
    full, rest = divmod(int(n_steps), spi)
    counts = [spi] * full + ([rest] if rest else [])
    ts = np.concatenate([[0.0], np.cumsum(counts) * DT])
    return ts, (spi if not rest else np.array(counts, dtype=np.int32))


def prepare(fw, designs, n_steps, spi=SPI):
    """Host side of a solve: design -> ControlParams -> flattened arrays -> device (dfx_set_params).  After this call the
    inputs are resident in HBM; it is NOT part of the timed region."""
    fw.timepoints, fw.step_counts = step_grid(n_steps, spi)
    sd = fw.solve_dynamics
    cps = [fw.control_params(d) for d in designs]
    flats = [sd._flatten(cp) for cp in cps]
    sd.engine.set_params(**{k: np.stack([f[k] for f in flats]) for k in flats[0]})
    sd._last = (cps, flats, fw.timepoints)


def execute(fw, obj, adjoint=True, spi=SPI):
    """The hot path on resident inputs: forward (+ objective + reverse sweep); returns device milliseconds + stats."""
    eng = fw.solve_dynamics.engine
    _, st_f = eng.forward(np.zeros((eng.batch, 2, eng.n_blocks, 3)), fw.timepoints, fw.step_counts, keep_trajectory=adjoint,
                          want_fields=False)
    out = {"fwd_ms": st_f["kernel_ms"], "fwd_launches": st_f["launches"], "streams": max(1, int(st_f.get("streams", 1))),
           "objective": None, "adj_ms": 0.0, "adj_launches": 0}
    if adjoint:
        out["objective"] = eng.objective_kinetic(obj.target_blocks)
        grads, st_a = eng.adjoint_kinetic(obj.target_blocks, which=("centroid_node_vectors", "void_angle0", "inertia"))
        out["adj_ms"], out["adj_launches"] = st_a["kernel_ms"], st_a["launches"]
        out["stage_checkpoint"] = bool(st_a.get("stage_checkpoint", 0))
        out["grad_norm"] = float(np.linalg.norm(grads["centroid_node_vectors"]))
    return out


def spin_up(fw, n_steps=500, spi=SPI):
    """Untimed burst on the resident inputs right before a timed region: the host-side preparation leaves the GPU idle
    for tens of milliseconds and the first launches after an idle period run at a lower clock."""
    eng = fw.solve_dynamics.engine
    ts = np.arange(n_steps // spi + 1) * (spi * DT)
    eng.forward(np.zeros((eng.batch, 2, eng.n_blocks, 3)), ts, spi, keep_trajectory=False, want_fields=False)


def run_once(fw, obj, designs, n_steps, adjoint=True, spi=SPI):
    prepare(fw, designs, n_steps, spi)
    if fw.solve_dynamics.engine.lib.dfx_device_count() > 0:
        spin_up(fw, spi=spi) if spi == SPI else None
    return execute(fw, obj, adjoint, spi)


def cpu_baseline(size, seed, budget_s=20.0):
    """The CPU port of the oracle (same algorithm, same tableau, OpenMP over blocks) on a bounded sample of the same
    workload: forward + adjoint of a few steps of the same lattice.  Thread count: the best of a short sweep."""
    import ctypes
    from oracle.cpu import load
    lib = load()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    fw, obj, designs = c3_problem(size, seed, 1, lib=lib)
    ncpu = os.cpu_count() or 1
    run_once(fw, obj, designs, 2, spi=2)                      # touch everything once
    best = (None, 0.0)
    for nt in sorted({1, min(8, ncpu), min(32, ncpu), min(96, ncpu)}):
        if gomp is not None:
            gomp.omp_set_num_threads(nt)
        elif nt != 1:
            continue
        prepare(fw, designs, 4, spi=2)
        t0 = time.perf_counter()
        execute(fw, obj, spi=2)
        rate = 4 / (time.perf_counter() - t0)
        if rate > best[1]:
            best = (nt, rate)
    nt, rate = best
    if gomp is not None:
        gomp.omp_set_num_threads(nt)
    n = int(max(4, min(2000, (budget_s * rate) // 2 * 2)))
    prepare(fw, designs, n, spi=2)               # same split as the GPU leg: inputs prepared outside the timed region
    t0 = time.perf_counter()
    execute(fw, obj, spi=2)
    dt = time.perf_counter() - t0
    return {"value": n * size * size / dt, "unit": "timesteps*units/s", "cores": nt, "kind": "port",
            "sample": f"{n} Dopri5 steps forward+adjoint of the same {size}x{size} lattice, 1 member "
                      f"(C++ port of the oracle, OpenMP, {nt} threads = best of a sweep on {ncpu} logical CPUs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--members", type=int, default=16,
                    help="independent designs per GPU integrated side by side (grid.y); capped so that the per-step "
                         "checkpoint of all members (72 B x units x steps each) fits the free HBM")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="member groups advanced concurrently, one HIP stream each")
    ap.add_argument("--no-single", action="store_true", help="skip the extra 1-member reference measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--all-ranks-device", type=int, default=-1, help="rehearsal only: put every rank on this device")
    args = ap.parse_args()
    if not os.path.exists(os.path.join(ROOT, "difflexmm_amd", "libdfx.so")):
        # build artefact missing (fresh checkout): compile it the way __graft_entry__.build() does, before any GPU call
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "difflexmm_amd", "csrc")], stdout=subprocess.DEVNULL)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    if args.all_ranks_device >= 0:
        local_rank = args.all_ranks_device
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = "cuda" if args.backend == "nccl" else "cpu"
    os.environ["DFX_STREAMS"] = str(args.streams)
    K = max(1, args.steps)                      # EXACTLY K steps are timed
    W = max(0, args.warmup)
    requested_members = args.members
    if not args.forward_only and args.backend == "nccl":
        # the reverse sweep reads a checkpoint of every step: 72 B per unit per step per member
        free_b, _ = torch.cuda.mem_get_info(local_rank)
        per_member = 72.0 * args.size * args.size * (max(K, W) + 1) + 64e6
        args.members = max(1, min(args.members, int(0.85 * free_b / per_member)))
        if dist is not None:                    # same work on every rank (weak scaling): take the smallest cap
            tm = torch.tensor([args.members], device=dev, dtype=torch.int64)
            dist.all_reduce(tm, op=dist.ReduceOp.MIN)
            args.members = int(tm.item())
        if args.members < args.streams:
            args.streams = args.members
            os.environ["DFX_STREAMS"] = str(args.streams)
        # Stage checkpoint (+120 B per unit and step: the reverse sweep then needs no recompute launches): the engine takes
        # it whenever it fits; decide here, by the same rule, so that the roofline leg below and the timed job run the
        # same kernels.
        if "DFX_STAGE_CHECKPOINT" not in os.environ:
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            need = (72.0 * (max(K, W) + 1) + 120.0 * max(K, W)) * args.size * args.size * args.members
            os.environ["DFX_STAGE_CHECKPOINT"] = "1" if need + 0.05 * total_b + 2e9 < free_b else "0"
    # (1) per-launch roofline of the dominant kernel: ONE stream, every launch integrates all `members` designs.
    #     This is the regime rocprofv3 can observe (its kernel trace serialises queues): `python bench.py --streams 1` under
    #     rocprofv3 --kernel-trace --stats reports the same average duration.  Measured on rank 0 BEFORE the timed job, on
    #     its own engine (closed again: HIP multiplexes streams onto few hardware queues, and the big checkpoint of the
    #     timed job is allocated afterwards).
    rr = None
    if rank == 0 and args.streams > 1:
        os.environ["DFX_STREAMS"] = "1"
        fwr, objr, desr = c3_problem(args.size, 3 + 1000 * rank, args.members, device=local_rank)
        os.environ["DFX_STREAMS"] = str(args.streams)
        Kr = min(K, 1000)
        fwr.solve_dynamics.engine.reserve(Kr, Kr // SPI + 2, keep_trajectory=not args.forward_only)
        run_once(fwr, objr, desr, SPI, adjoint=not args.forward_only)
        if Kr % SPI:
            run_once(fwr, objr, desr, Kr % SPI, adjoint=not args.forward_only)
        torch.cuda.synchronize()
        rr = run_once(fwr, objr, desr, Kr, adjoint=not args.forward_only)
        fwr.solve_dynamics.engine.close()
        del fwr, objr
    single = None
    if rank == 0 and world == 1 and args.members > 1 and not args.forward_only and not args.no_single:
        # the same config with ONE design per GPU (launch-bound: one wave per SIMD), for reference; measured before the
        # timed job (the first solves after releasing a > 200 GB checkpoint were seen to run at half speed)
        fw1, obj1, des1 = c3_problem(args.size, 3, 1, device=local_rank)
        K1 = min(K, 2500)
        fw1.solve_dynamics.engine.reserve(K1, K1 // SPI + 2, keep_trajectory=True)
        run_once(fw1, obj1, des1, SPI)
        if K1 % SPI:
            run_once(fw1, obj1, des1, K1 % SPI)
        prepare(fw1, des1, K1)
        spin_up(fw1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r1 = execute(fw1, obj1)
        torch.cuda.synchronize()
        w1 = time.perf_counter() - t1
        single = {"members_per_gpu": 1, "steps": K1, "value": K1 * args.size * args.size / w1,
                  "forward_only_value": K1 * args.size * args.size / (r1["fwd_ms"] * 1e-3),
                  "fwd_launch_us": 1e3 * r1["fwd_ms"] / max(1, r1["fwd_launches"]),
                  "stage_checkpoint": bool(r1.get("stage_checkpoint", False)),
                  "device_ms": {"forward": r1["fwd_ms"], "adjoint": r1["adj_ms"]}}
        fw1.solve_dynamics.engine.close()
        del fw1, obj1
    fw, obj, designs = c3_problem(args.size, 3 + 1000 * rank, args.members, device=local_rank)
    fw.solve_dynamics.engine.reserve(max(K, W), max(K, W) // SPI + 2, keep_trajectory=not args.forward_only)
    if W:
        run_once(fw, obj, designs, W, adjoint=not args.forward_only)
    # hipGraphs are instantiated on first use: make sure every segment length the K timed steps replay has been used once
    # (a full output interval and the shorter last interval), whatever W was
    if K >= SPI and (W < SPI or W % SPI):
        run_once(fw, obj, designs, SPI, adjoint=not args.forward_only)
    if K % SPI:
        run_once(fw, obj, designs, K % SPI, adjoint=not args.forward_only)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    tp0 = time.perf_counter()
    prepare(fw, designs, K)                    # inputs resident in HBM before the timed region
    host_prepare_ms = 1e3 * (time.perf_counter() - tp0)   # design -> ControlParams -> packed arrays -> H2D, reported, not timed
    spin_up(fw)
    barrier()
    t0 = time.perf_counter()
    res = execute(fw, obj, adjoint=not args.forward_only)
    barrier()
    wall = time.perf_counter() - t0
    objective = res["objective"] if res["objective"] is not None else np.zeros(args.members)
    if dist is not None:
        tw = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        mine = torch.tensor(np.asarray(objective, dtype=np.float64), device=dev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)          # the single collective of the path: objectives over xGMI
        objective = torch.cat(gathered).cpu().numpy()
    if rank == 0:
        n_units = args.size * args.size
        total_units_steps = K * n_units * args.members * world
        streams = res["streams"]

        def per_launch(r, n_streams):
            """(fwd launch us, adj launch us): region device time / launches issued per stream."""
            f_us = 1e3 * r["fwd_ms"] / max(1.0, r["fwd_launches"] / n_streams)
            a_us = None
            if r["adj_launches"] and r.get("stage_checkpoint"):
                a_us = 1e3 * r["adj_ms"] / (r["adj_launches"] / n_streams)     # reverse stages only (stage checkpoint kept)
            elif r["adj_launches"]:
                n_adj = r["adj_launches"] * 6.0 / 11.0 / n_streams   # per reverse step: 5 recomputed forward stages + 6 reverse stages
                a_us = max(1e-9, (1e3 * r["adj_ms"] - n_adj * (5.0 / 6.0) * f_us) / n_adj)
            return f_us, a_us

        fw.solve_dynamics.engine.close()
        if rr is None:
            rr = res
        fwd_us, adj_us = per_launch(rr, 1)
        roof_bytes = BYTES_FWD_STAGE * n_units * args.members
        achieved = roof_bytes / (fwd_us * 1e-6) / 1e9
        line = {
            "metric": "timesteps*rigid-units/s (forward + adjoint)" if not args.forward_only else "timesteps*rigid-units/s (forward)",
            "value": total_units_steps / wall, "unit": "timesteps*units/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * wall / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3: {args.size}x{args.size} quads, nonlinear ligaments + damping + angle contact, "
                                   f"pulse drive, fixed-step Dopri5 dt={DT:.3e}s, {K} of 50000 steps, "
                                   f"{'forward only' if args.forward_only else 'forward + adjoint wrt 66048 geometry params'}",
                       "members_per_gpu": args.members, "members_requested": requested_members, "concurrent_streams": streams,
                       "stage_checkpoint": bool(res.get("stage_checkpoint", False)), "integrator": "dopri5-fixed",
                       "steps_per_output": SPI},
            "forward_only_value": K * n_units * args.members * world / (res["fwd_ms"] * 1e-3),
            "device_ms": {"forward": res["fwd_ms"], "adjoint": res["adj_ms"], "wall": 1e3 * wall},
            "host_prepare_ms": host_prepare_ms, "value_with_host_prepare": total_units_steps / (wall + 1e-3 * host_prepare_ms),
            "launches": {"forward": res["fwd_launches"], "adjoint": res["adj_launches"]},
            "objective": [float(x) for x in np.atleast_1d(objective)][:8],
            "roofline": {"bound": "hbm", "kernel": "k_fwd_stage<nonlinear,contact>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": load_pmc_traffic(args.members),
                         "bytes_per_launch": roof_bytes, "launch_us": fwd_us, "members_per_launch": args.members,
                         "measured_with": "1 stream, HIP events around the forward region / launches"},
        }
        if adj_us:
            adj_bytes = BYTES_ADJ_STAGE_REBUILD if rr.get("stage_checkpoint") else BYTES_ADJ_STAGE
            a2 = adj_bytes * n_units * args.members / (adj_us * 1e-6) / 1e9
            line["roofline_adjoint_kernel"] = {"kernel": "k_adj_stage<nonlinear,contact>", "achieved": a2, "peak": HBM_PEAK_GBS,
                                               "unit": "GB/s", "frac": a2 / HBM_PEAK_GBS, "launch_us": adj_us,
                                               "bytes_per_launch": adj_bytes * n_units * args.members,
                                               "rebuilds_stage_records": bool(rr.get("stage_checkpoint"))}
        if streams > 1:
            # (2) the timed job itself: `streams` member groups overlap on the chip; aggregate algorithmic bytes / region time
            f_eff, a_eff = per_launch(res, streams)
            agg = {"concurrent_streams": streams, "members_per_launch": args.members / streams,
                   "fwd_stage_period_us": f_eff, "fwd_achieved": roof_bytes / (f_eff * 1e-6) / 1e9}
            agg["fwd_frac"] = agg["fwd_achieved"] / HBM_PEAK_GBS
            if a_eff:
                agg["adj_stage_period_us"] = a_eff
                agg["adj_achieved"] = (BYTES_ADJ_STAGE_REBUILD if res.get("stage_checkpoint") else BYTES_ADJ_STAGE) * n_units * args.members / (a_eff * 1e-6) / 1e9
                agg["adj_frac"] = agg["adj_achieved"] / HBM_PEAK_GBS
            line["roofline_concurrent"] = agg
        if single is not None:
            line["single_system"] = single
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.size, 3)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def load_pmc_traffic(members_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json:
    bytes per member and launch, with the guide's x2 correction of FETCH_SIZE), or null."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            return d.get("k_fwd_stage_bytes_per_member_launch") * members_per_launch
        except Exception:
            return None
    return None


if __name__ == "__main__":
    main()
