#!/usr/bin/env python3
"""BASELINE config 5: the multi-input focusing inverse design (problems/quads_focusing_multi_input.py) as an ensemble of
independent optimisations, sharded over GPUs.

    python examples/multi_input_ensemble.py --members 32 --iterations 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/multi_input_ensemble.py --members 256 --iterations 20      # any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE

24x16 quads, three inputs (left / right / bottom edge, input shifts 0 / -2 / -4), one objective per design = weighted sum
of the three target kinetic energies; every member starts from its own perturbed design (seeds 1000, 1001, ...) and runs
the reference's loop (method of moving asymptotes under the angle / edge-length constraints).  One rank = one GPU = one
contiguous chunk of members integrated side by side (grid.y of every launch); members advance in lock-step so each round
is three batched forward + reverse sweeps; no data-path collective, the final objectives are combined with one RCCL
all-gather inside libdfx (difflexmm_amd/ensemble.py: RcclComm -> dfx_gather_objectives).  No PyTorch in the ranks.
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench_extras(objective, steps, nb, mine, world, args):
    """`roofline` / `cpu_baseline` / `csrc` of the bench line (bench.py --workload c5): device time per Runge-Kutta stage of the LAST
    evaluation of input 0's engine (HIP events around its sweeps / stages) against the 8 TB/s roofline in algorithmic bytes (SURVEY
    8(d): 344 B forward, 624 B reverse per unit and stage, quads + contact), and the C++ port of the oracle on the same 24x16 lattice
    (one member, one input, bounded sample) on this host."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sd = objective.objectives[0].forward.solve_dynamics
    st, sa = sd.stats, sd.adjoint_stats
    f_us, a_us = 1e3 * st["kernel_ms"] / (steps * 6), 1e3 * sa["kernel_ms"] / (steps * 6)
    builds = {"forward": bench.BUILD_NAMES.get(int(st.get("tile_kernels", 0))), "adjoint": bench.BUILD_NAMES.get(int(sa.get("tile_kernels", 0)))}
    out = {"roofline": bench.stage_roofline("reverse stage <nonlinear,contact>", bench.BYTES_ADJ_STAGE, nb * mine, a_us, kernels=builds,
                                            members_per_stage=mine, measured_with="HIP events around the reverse sweep of input 0's last "
                                            "evaluation / (steps x 6 stages), rank 0"),
           "roofline_forward_kernel": bench.stage_roofline("forward stage <nonlinear,contact>", bench.BYTES_FWD_STAGE, nb * mine, f_us),
           "csrc": bench.source_ids()}
    if world == 1 and not args.cpu_port:
        def make(lib):
            from difflexmm_amd import problems as P
            fw = objective.objectives[0].forward
            fwc = P.QuadsFocusingForward(**{**{k: getattr(fw, k) for k in ("n1_blocks", "n2_blocks", "spacing", "bond_length", "k_stretch", "k_shear",
                                                "k_rot", "density", "damping", "amplitude", "loading_rate", "input_delay", "n_excited_blocks",
                                                "loaded_side", "input_shift", "simulation_time", "n_timepoints", "use_contact", "k_contact",
                                                "min_angle", "cutoff_angle")}, "steps_per_interval": 50, "batch": 1, "_lib": lib})
            fwc.setup()
            objc = P.TargetKineticEnergy(fwc, (2, 2), (args.n1 // 6, args.n2 // 5))
            return fwc, objc, [fwc.geometry.get_design_from_rotated_square(25 * math.pi / 180)]
        out["cpu_baseline"] = bench.cpu_baseline(0, 0, n_steps=1000, repeats=3, budget_s=30.0, make=make,
                                                 what=f"the same {args.n1}x{args.n2} lattice, one input")
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=32, help="designs in the whole ensemble")
    ap.add_argument("--iterations", type=int, default=10, help="objective evaluations per design (nlopt maxeval)")
    ap.add_argument("--n1", type=int, default=24)
    ap.add_argument("--n2", type=int, default=16)
    ap.add_argument("--steps-per-interval", type=int, default=100)
    ap.add_argument("--timepoints", type=int, default=41)
    ap.add_argument("--backend", default="rccl", help="rccl (inside libdfx) | socket (plain TCP: rehearsals without one GPU per rank)")
    ap.add_argument("--host-workers", type=int, default=-1,
                    help="host processes for the members' constraint evaluations and MMA sub-problems (0: in this process; "
                         "-1: min(32, CPUs of this rank - 1))")
    ap.add_argument("--cpu-port", action="store_true", help="use the oracle's CPU port instead of libdfx (rehearsal without a GPU)")
    ap.add_argument("--pipeline", action="store_true",
                    help="the two halves of the chunk take turns on the device, the workers run one half's MMA steps while the other half is "
                         "integrated (same iterates; measured on config 5: no gain, half batches run the device less efficiently)")
    ap.add_argument("--all-ranks-device", type=int, default=-1, help="rehearsal only: put every rank on this device")
    ap.add_argument("--json", action="store_true", help="rank 0 also prints one JSON line in the format of bench.py (bench.py --workload c5)")
    args = ap.parse_args(argv)

    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if args.all_ranks_device >= 0:
        local_rank = args.all_ranks_device
        os.environ["DFX_PERSIST"] = "0"      # several processes on one GPU: one launch per stage (bench.py says why)
    from difflexmm_amd import problems as P
    from difflexmm_amd import ensemble
    from difflexmm_amd.optimize import MemberWorkers
    # the host workers are forked FIRST: before RCCL / the engines initialise the GPU in this process
    n_workers = args.host_workers if args.host_workers >= 0 else min(32, max(1, len(os.sched_getaffinity(0)) // max(1, world) - 1))
    workers = MemberWorkers(n_workers) if n_workers > 0 else None
    from difflexmm_amd.ensemble import gather_objectives, shard_bounds
    comm = ensemble.init_from_env(args.backend, device=local_rank) if world > 1 else ensemble.SerialComm()
    lib = None
    if args.cpu_port:
        from oracle.cpu import load
        lib = load()

    lo, hi = shard_bounds(args.members, rank, world)
    mine = hi - lo
    pipeline = workers is not None and args.pipeline and mine >= 2 and mine % 2 == 0
    per_call = mine // 2 if pipeline else mine
    spacing, bond, rho, ksh, kr, freq = 15.0, 2.25, 6.18e-9, 1.19, 1.5, 30.0
    nb = args.n1 * args.n2
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh)] * 2
                                + [2 * math.sqrt(0.02175026 * rho * spacing ** 4 * kr)]) * np.ones((nb, 1))
    forwards = []
    for side, shift in (("left", 0), ("right", -2), ("bottom", -4)):
        fw = P.QuadsFocusingForward(
            n1_blocks=args.n1, n2_blocks=args.n2, spacing=spacing, bond_length=bond, k_stretch=120.0, k_shear=ksh, k_rot=kr,
            density=rho, damping=damping, amplitude=7.5, loading_rate=freq, input_delay=0.1 / freq, n_excited_blocks=2,
            loaded_side=side, input_shift=shift, simulation_time=2.0 / freq, n_timepoints=args.timepoints,
            use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180,
            steps_per_interval=args.steps_per_interval, batch=per_call, device=local_rank, _lib=lib)
        fw.setup()
        forwards.append(fw)
    objective = P.MultiInputTargetKineticEnergy(forwards, (2, 2), (args.n1 // 6, args.n2 // 5), weights=[1.0, 1.0, 1.0])
    base = forwards[0].geometry.get_design_from_rotated_square(25 * math.pi / 180)
    x0s = []
    for m in range(lo, hi):
        rng = np.random.default_rng(1000 + m)
        x0s.append(tuple(b + rng.uniform(-0.02 * spacing, 0.02 * spacing, b.shape) for b in base))
    amin = 5 * math.pi / 180
    t0 = time.perf_counter()
    best, logs = P.run_ensemble_optimization(objective, x0s, args.iterations, lower_bound=-0.3 * spacing, upper_bound=0.3 * spacing,
                                             min_void_angle=amin, min_block_angle=amin, min_edge_length=0.1 * spacing,
                                             verbose=(rank == 0), workers=workers, pipeline=pipeline)
    wall = time.perf_counter() - t0
    first = gather_objectives([l["objective_values"][0] for l in logs], args.members)
    final = gather_objectives([l["mma"].fun for l in logs], args.members)
    if rank == 0:
        solves = 3 * args.members * args.iterations
        steps = (args.timepoints - 1) * args.steps_per_interval
        print(f"{args.members} designs x 3 inputs x {args.iterations} evaluations on {world} rank(s){' (two halves pipelined)' if pipeline else ''}: {wall:.1f} s, "
              f"{solves / wall:.1f} forward+adjoint solves/s, {solves * steps * nb / wall:.3e} timesteps*units/s")
        ev = logs[0]["evaluation_seconds"]
        print("evaluations of the whole ensemble (forward + reverse sweeps of the three inputs + design maps), s: "
              + " ".join(f"{t:.2f}" for t in ev) + "   (the first allocates the engines' checkpoints)")
        dev = sum(getattr(o, "device_ms", 0.0) for o in objective.objectives) * 1e-3
        print(f"device time of the forward + reverse sweeps, summed over the three engines: {dev:.1f} s (wall {wall:.1f} s; the engines of "
              f"the three inputs overlap when each runs a single stream); the rest is host work per round: design -> ControlParams "
              f"-> packed arrays, gradient maps back to the design, the MMA sub-problems ({n_workers} host worker processes)")
        print("objective, first evaluation :", np.array2string(first, precision=3, max_line_width=160))
        print("objective, best feasible    :", np.array2string(final, precision=3, max_line_width=160))
        if args.json:
            import json
            info = comm.info() if hasattr(comm, "info") else {}
            print(json.dumps({
                "metric": "timesteps*rigid-units/s (forward + adjoint, inside the full optimisation loop)", "value": solves * steps * nb / wall,
                "unit": "timesteps*units/s", "n_gpus": world, "steps": args.iterations, "warmup": 0, "ms_per_step": 1e3 * wall / args.iterations,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"C5: problems/quads_focusing_multi_input, {args.members} designs (seeds 1000..) x 3 inputs, {args.n1}x{args.n2} "
                                       f"quads, {steps} Dopri5 steps per solve, {args.iterations} objective evaluations per design: MMA under "
                                       "the angle / edge-length constraints, all designs in lock-step, one all-gather of objectives",
                           "designs_total": args.members, "members_per_gpu": mine, "host_workers": n_workers,
                           "collective": info.get("collective", "none (1 rank)" if world == 1 else args.backend),
                           "ranks_seen": info.get("ranks_seen", world), "rccl_version": info.get("rccl_runtime")},
                "solves_per_s": solves / wall, "wall_s": wall, "device_s_rank0": dev, "evaluation_seconds_rank0": [float(t) for t in ev],
                # the first lock-step evaluation allocates the engines' checkpoints (a fresh process pays it once per run); the others are
                # what an optimisation of hundreds of iterations runs at
                "first_evaluation_s": float(ev[0]) if ev else None,
                "steady_state": None if len(ev) < 2 else {
                    "evaluation_s": float(np.median(ev[1:])),
                    "value": 3 * args.members * steps * nb / float(np.median(ev[1:])), "unit": "timesteps*units/s",
                    "note": "median of evaluations 2.. of the whole ensemble on rank 0 (forward + reverse sweeps of the three inputs + design maps); "
                            "`value` above also carries the first (allocating) evaluation and the MMA sub-problems between evaluations"},
                "objective_first": [float(x) for x in first[:8]], "objective_best": [float(x) for x in final[:8]],
                "objectives_gathered": int(len(final)), **bench_extras(objective, steps, nb, mine, world, args)}), flush=True)
    if workers is not None:
        workers.close()
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
