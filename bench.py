#!/usr/bin/env python3
"""bench.py -- timesteps x rigid-units / s of the hot path (forward + discrete adjoint) on MI355X.

Workload (BASELINE.json configs[2], SURVEY 8(d) "C3"): 128x128 quad lattice (16 384 rigid units), nonlinear
ligaments + viscous damping + angle-based contact, raised-cosine displacement pulse on 2 left-edge blocks,
clamped corners, Dormand-Prince tableau on a fixed grid with dt = (2/f)/50 000, one output every 250 steps,
objective = kinetic energy of 2x2 target blocks, ControlParams-shaped gradient (node vectors, void angles, inertia of every design).
One "step" = one RK step (6 RHS evaluations) of every member, forward AND reverse.  `--steps K` times exactly K
steps (output intervals of 250 steps and, if K is not a multiple of 250, one shorter last interval); the full config is
K = 50 000.  16 independent designs per GPU while the state checkpoint AND the stage checkpoint (72 + 120 B per unit and
step: the reverse sweep then needs no recompute launches) fit the 288 GB; longer runs fall back to the state checkpoint
alone and then to fewer members.

Two deviations from the C3 text, both so that a SHORT timed window differentiates something (round-1 verdict: with the
paper's input delay of 0.1/f = 3.3 ms and the target 21 x 25 blocks away, the first 20 steps -- 27 us -- integrate a lattice
at rest and the reverse sweep propagates exact zeros): the pulse starts at t = 0 (`--input-delay`), and the 2x2 target sits
next to the driven blocks (`--target-shift`; the C3 placement is `--target-shift 21 25`).  Neither changes a launch or a byte.

What the timed region is (round 4): ONE library call, `dfx_forward_kinetic_value_and_grad` -- forward solve, objective, reverse
sweep, as `jit(value_and_grad(objective))` is one program in the reference -- on inputs resident in HBM, with the gradients left in
HBM too (device pointers come back; `--outputs host` / `DFX_BENCH_FUSED=0` restore rounds 1-3's region: two calls, gradients copied
to pinned host memory inside it).  The PCIe legs either side are measured and reported next to `value`, never inside it:
`host_prepare_ms` (design -> packed parameters -> H2D) and `outputs.value_with_outputs_on_host`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--members M] [--size 128]

N > 1: one rank per GPU, every rank integrates its own designs (weak scaling, no data-path collective), objectives are
gathered with ONE RCCL all-gather inside libdfx.  Launched either by the driver (`python -m torch.distributed.run ...
bench.py --gpus N`: the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) or, when WORLD_SIZE is not set, by this
script itself: it starts N rank processes BEFORE touching the GPU and relays rank 0's line.
Host code is Python + ctypes + NumPy: no PyTorch.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPI = 250                     # steps between outputs: 50 000 steps / 200 output intervals
FREQ = 30.0
DT = (2.0 / FREQ) / 50000.0
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md, chip-level parameters
FP64_VALU_PEAK_TFLOPS = 78.6  # vector fp64: 256 CUs x 64 FMA lanes per clock x 2 flops x 2.4 GHz (AMD: 78.6); no matrix cores on this path
# ALGORITHMIC bytes per rigid unit per launch
BYTES_FWD_STAGE = 272 + 72          # one RHS evaluation (quads + contact) + its share of the stage combine (432 / 6)
BYTES_ADJ_STAGE = 272 + 96 + 256    # stage data + lambda / Ybar read-write + parameter-gradient read-modify-write = 624
ROOFLINE_LEG_STEPS = 250            # length of the per-launch measurement (1 stream, all members per launch), whatever K is
# the same accounting for the other lattices of BASELINE.json (RHS bytes 256 quads without contact, 244 kagome + contact;
# reverse: + 96 lambda / Ybar + read-modify-write of 2 x (16 n_npb + 16 + 24 + 24) parameter-gradient bytes)
BYTES_FWD_STAGE_C2 = 256 + 72
BYTES_FWD_STAGE_KAGOME = 244 + 72
BYTES_ADJ_STAGE_KAGOME = 244 + 96 + 2 * (48 + 16 + 24 + 24)
BUILD_NAMES = {0: "stage launches (generic builds)", 1: "tile kernels", 2: "stage launches (per-stage builds)", 3: "persistent stage loop"}


def c3_problem(size, seed, members, lib=None, device=0, input_delay=0.0, target_shift=None, contact_cutoff_deg=-10.0, contact_min_deg=-15.0):
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    spacing, bond = 15.0, 2.25
    rho, ksh, kr = 6.18e-9, 1.19, 1.5
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh),
                                 2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh),
                                 2 * math.sqrt(0.02175026 * rho * spacing ** 4 * kr)]) * np.ones((size * size, 1))
    fw = QuadsFocusingForward(
        n1_blocks=size, n2_blocks=size, spacing=spacing, bond_length=bond, k_stretch=120.0, k_shear=ksh, k_rot=kr,
        density=rho, damping=damping, amplitude=7.5, loading_rate=FREQ, input_delay=input_delay, n_excited_blocks=2,
        loaded_side="left", input_shift=0, simulation_time=2.0 / FREQ, n_timepoints=201, use_contact=True,
        k_contact=1.5, min_angle=contact_min_deg * math.pi / 180, cutoff_angle=contact_cutoff_deg * math.pi / 180,
        steps_per_interval=SPI,
        batch=members, device=device, _lib=lib)
    fw.setup()
    if target_shift is None:
        target_shift = (1 - (size - 2) // 2, 0)        # columns 1-2 of the two driven rows: next to the driven blocks
    obj = TargetKineticEnergy(fw, (2, 2), tuple(target_shift))
    designs = []
    for m in range(members):
        rng = np.random.default_rng(seed + m)
        base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
        designs.append(tuple(b + rng.uniform(-0.02 * spacing, 0.02 * spacing, b.shape) for b in base))
    return fw, obj, designs


def step_grid(n_steps, spi=SPI, t_start=0.0):
    """EXACTLY n_steps RK steps of size DT: full output intervals of `spi` steps and, when n_steps is not a multiple of
    spi, one shorter last interval.  Returns (timepoints, steps per interval)."""
    full, rest = divmod(int(n_steps), spi)
    counts = [spi] * full + ([rest] if rest else [])
    ts = t_start + np.concatenate([[0.0], np.cumsum(counts) * DT])
    return ts, (spi if not rest else np.array(counts, dtype=np.int32))


def prepare(fw, designs, n_steps, spi=SPI, t_start=0.0):
    """Host side of a solve: design -> ControlParams -> flattened arrays -> device (dfx_set_params).  After this call the
    inputs are resident in HBM; it is NOT part of the timed region."""
    fw.timepoints, fw.step_counts = step_grid(n_steps, spi, t_start)
    sd = fw.solve_dynamics
    if isinstance(designs, list) and len(designs) > 1:
        from difflexmm_amd.problems import prefetch_designs
        prefetch_designs(fw, designs)              # the geometry of all designs in one native pass (no-op when cached)
    cps, flats = sd.prepare([fw.control_params(d) for d in designs])
    sd._last = (cps, flats, fw.timepoints)


FUSED_CALL = os.environ.get("DFX_BENCH_FUSED", "1") != "0"      # (A/B: the two-call sequence of rounds 1-3)


def execute(fw, obj, adjoint=True, spi=SPI, device_outputs=False):
    """The hot path on resident inputs: forward (members start at rest: no upload) + objective + reverse sweep; returns device
    milliseconds + stats.  Gradients: views of the engine's pinned result area (they crossed PCIe inside this call), or with
    ``device_outputs`` DeviceArray handles to the accumulators in HBM (dfx_kinetic_value_and_grad_device: outputs resident like the
    inputs, what the reference's jit(value_and_grad) hands back) -- `fetch` downloads them afterwards."""
    eng = fw.solve_dynamics.engine
    which = ("centroid_node_vectors", "void_angle0", "inertia")
    fused = adjoint and FUSED_CALL and hasattr(eng.lib, "dfx_forward_kinetic_value_and_grad") and np.ndim(fw.timepoints) == 1
    if fused:      # one library call, as jit(value_and_grad(objective)) is one program: the host does not wait between the two sweeps
        objective, grads, st_f, st_a = eng.forward_kinetic_value_and_grad(None, fw.timepoints, fw.step_counts, obj.target_blocks,
                                                                          which=which, device=device_outputs)
    else:
        _, st_f = eng.forward(None, fw.timepoints, fw.step_counts, keep_trajectory=adjoint, want_fields=False)
    out = {"fwd_ms": st_f["kernel_ms"], "fwd_launches": st_f["launches"], "streams": max(1, int(st_f.get("streams", 1))),
           "objective": None, "adj_ms": 0.0, "adj_launches": 0, "fused_call": bool(fused), "fwd_build": int(st_f.get("tile_kernels", 0)),
           "steps": int(st_f.get("steps", 0))}
    if adjoint:
        if fused:
            out["objective"] = objective
        else:
            out["objective"], grads, st_a = eng.kinetic_value_and_grad(obj.target_blocks, which=which, device=device_outputs)
        out["adj_ms"], out["adj_launches"] = st_a["kernel_ms"], st_a["launches"]
        out["adj_build"] = int(st_a.get("tile_kernels", 0))
        out["stage_checkpoint"] = bool(st_a.get("stage_checkpoint", 0))
        out["checkpoint"] = {1: "records", 2: "segments"}.get(st_a.get("checkpoint_records", 0)) or ("stages" if st_a.get("stage_checkpoint", 0) else "state")
        out["grads"] = grads
    return out


def fetch(res):
    """Download device-resident gradients of `execute(device_outputs=True)`; returns the seconds it took (0 for host views)."""
    g = res.get("grads")
    if not g or not any(hasattr(a, "to_host") for a in g.values()):
        return 0.0
    t0 = time.perf_counter()
    res["grads"] = {k: (a.to_host() if hasattr(a, "to_host") else a) for k, a in g.items()}
    return time.perf_counter() - t0


def grad_norm(res):
    fetch(res)
    g = res.get("grads")
    return None if g is None else float(np.sqrt(sum(float(np.vdot(a, a)) for a in g.values())))


def spin_up(fw, n_steps=500, spi=SPI):
    """Untimed burst on the resident inputs right before a timed region: the host-side preparation leaves the GPU idle
    for tens of milliseconds and the first launches after an idle period run at a lower clock."""
    eng = fw.solve_dynamics.engine
    ts = np.arange(n_steps // spi + 1) * (spi * DT)
    eng.forward(None, ts, spi, keep_trajectory=False, want_fields=False)


def run_once(fw, obj, designs, n_steps, adjoint=True, spi=SPI, device_outputs=False):
    prepare(fw, designs, n_steps, spi)
    if fw.solve_dynamics.engine.lib.dfx_device_count() > 0 and spi == SPI:
        spin_up(fw, spi=spi)
    return execute(fw, obj, adjoint, spi, device_outputs)


def stage_roofline(kernel, bytes_per_unit, units, stage_us, **extra):
    """Roofline entry of one Runge-Kutta stage of `units` rigid units that took `stage_us` on the device (HIP events around the sweep /
    stages): ALGORITHMIC bytes against the 8 TB/s HBM peak -- a launch, or one stage of the persistent loop."""
    ach = bytes_per_unit * units / (stage_us * 1e-6) / 1e9
    d = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
         "algorithmic_bytes_per_unit": bytes_per_unit, "units_per_stage": units, "stage_us": stage_us}
    d.update(extra)
    return d


def c2_problem(members, lib=None, device=0, size=32):
    """BASELINE configs[1] (SURVEY 8(d) "C2"): 32x32 quads, nonlinear ligaments + damping, NO contact, seed 2, fixed-step Dopri5
    with dt = (2/f)/10 000, forward only.  The caller is the reference's own: problems/quads_focusing.py restated."""
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    spacing, bond, rho, ksh, kr = 15.0, 2.25, 6.18e-9, 1.19, 1.5
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh)] * 2 +
                                [2 * math.sqrt(0.02175026 * rho * spacing ** 4 * kr)]) * np.ones((size * size, 1))
    fw = QuadsFocusingForward(n1_blocks=size, n2_blocks=size, spacing=spacing, bond_length=bond, k_stretch=120.0, k_shear=ksh, k_rot=kr,
                              density=rho, damping=damping, amplitude=7.5, loading_rate=FREQ, input_delay=0.0, n_excited_blocks=2,
                              loaded_side="left", input_shift=0, simulation_time=2.0 / FREQ, n_timepoints=41, use_contact=False,
                              steps_per_interval=SPI, batch=members, device=device, _lib=lib)
    fw.setup()
    obj = TargetKineticEnergy(fw, (2, 2), (size // 6, size // 5))
    designs = []
    for m in range(members):
        rng = np.random.default_rng(2 + m)
        base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
        designs.append(tuple(b + rng.uniform(-0.02 * spacing, 0.02 * spacing, b.shape) for b in base))
    return fw, obj, designs


C2_DT = (2.0 / FREQ) / 10000.0


def c2_solve(fw, designs, n_steps, sync):
    """Forward-only solve of `n_steps` steps of size C2_DT on resident inputs; returns (wall s, stats)."""
    spi = min(SPI, n_steps)
    counts = [spi] * (n_steps // spi) + ([n_steps % spi] if n_steps % spi else [])
    fw.timepoints = np.concatenate([[0.0], np.cumsum(counts) * C2_DT])
    fw.step_counts = spi if not n_steps % spi else np.array(counts, dtype=np.int32)
    sd = fw.solve_dynamics
    if isinstance(designs, list) and len(designs) > 1:
        from difflexmm_amd.problems import prefetch_designs
        prefetch_designs(fw, designs)              # the geometry of all designs in one native pass (no-op when cached)
    cps, flats = sd.prepare([fw.control_params(d) for d in designs])
    sd._last = (cps, flats, fw.timepoints)
    sync()
    t0 = time.perf_counter()
    _, st = sd.engine.forward(None, fw.timepoints, fw.step_counts, keep_trajectory=False, want_fields=False)
    sync()
    return time.perf_counter() - t0, st


def launch_bound_block(device, sync, single, c4_steps=400, c2_steps=2000):
    """The regimes whose launches do NOT fill the chip, as the driver's default line sees them (round-4 verdict): BASELINE configs[1]
    with ONE member (32x32 quads, 64 waves in all), C3 with one member (one wave per SIMD: `single_system`), config 4 at its per-GPU
    width of an 8-GPU run (8 designs of the 64x64-cell kagome lattice).  Device time per Runge-Kutta stage and its fraction of the
    8 TB/s roofline in algorithmic bytes; `kernels` says what ran (the persistent stage loop where a solve fits on the chip at once)."""
    out = {}
    # (the engine chooses streams and checkpoint level itself here, as for any caller: the C3 legs of this script pin both)
    pinned = {k: os.environ.pop(k) for k in ("DFX_STREAMS", "DFX_CHECKPOINT") if k in os.environ}
    try:
        _launch_bound_legs(out, device, sync, single, c4_steps, c2_steps)
    finally:
        os.environ.update(pinned)
    return out


def _launch_bound_legs(out, device, sync, single, c4_steps, c2_steps):
    fw, obj, designs = c2_problem(1, device=device)
    c2_solve(fw, designs, 500, sync)
    wall, st = c2_solve(fw, designs, c2_steps, sync)
    us = 1e3 * st["kernel_ms"] / (c2_steps * 6)
    out["c2_1_member"] = {"lattice": "32x32 quads, nonlinear + damping, forward only", "steps": c2_steps, "stage_us": us,
                          "value": c2_steps * 1024 / wall, "device_value": c2_steps * 1024 / (st["kernel_ms"] * 1e-3),
                          "frac": BYTES_FWD_STAGE_C2 * 1024 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernels": BUILD_NAMES.get(int(st.get("tile_kernels", 0)))}
    fw.solve_dynamics.engine.close()
    if single is not None:
        out["c3_1_member"] = {"lattice": "128x128 quads + contact, forward + adjoint", "steps": single["steps"],
                              "fwd_stage_us": single["fwd_stage_us"], "adj_stage_us": single["adj_stage_us"], "value": single["value"],
                              "fwd_frac": single["roofline"]["forward_frac"], "adj_frac": single["roofline"]["frac"], "kernels": single["kernels"]}
    if single is not None:
        # ... and the same single design over the WHOLE horizon of C3 as written (50 000 steps, pulse delayed by 0.1/f, target at (21, 25)):
        # what a user of the reference runs.  Its 275 GB of stage records do not fit next to everything else, so the engine
        # takes the segments level (every output interval integrated twice, both sweeps in the persistent loop)
        t_d = 0.1 / FREQ
        fwf, objf, desf = c3_problem(128, 3, 1, device=device, input_delay=t_d, target_shift=(128 // 6, 128 // 5))
        fwf.solve_dynamics.engine.reserve(50000, 50000 // SPI + 2, keep_trajectory=True)
        prepare(fwf, desf, 2 * SPI, t_start=t_d)
        execute(fwf, objf)
        prepare(fwf, desf, 50000)
        sync()
        tf = time.perf_counter()
        rf = execute(fwf, objf)
        sync()
        wf = time.perf_counter() - tf
        out["c3_1_member_whole_horizon"] = {"lattice": "128x128 quads + contact, ONE design, 50 000 steps as C3 is written, forward + adjoint",
                                            "value": 50000 * 16384 / wf, "wall_s": wf, "checkpoint": rf.get("checkpoint"),
                                            "device_ms": {"forward": rf["fwd_ms"], "adjoint": rf["adj_ms"]},
                                            "kernels": {"forward": BUILD_NAMES.get(rf.get("fwd_build")), "adjoint": BUILD_NAMES.get(rf.get("adj_build"))},
                                            "objective": float(np.atleast_1d(rf["objective"])[0]), "grad_norm": grad_norm(rf)}
        fwf.solve_dynamics.engine.close()
    fw4, obj4, K4 = c4_problem(8, c4_steps, device=device)
    designs4 = []
    for seed in range(100, 108):
        rng = np.random.default_rng(seed)
        designs4.append(tuple(rng.uniform(-0.3, 0.3, sh) for sh in fw4.geometry.design_shapes()))
    obj4.value_and_grad(designs4)
    sync()
    t0 = time.perf_counter()
    obj4.value_and_grad(designs4)
    sync()
    wall4 = time.perf_counter() - t0
    sd = fw4.solve_dynamics
    n4 = fw4.geometry.n_blocks
    f_us, a_us = 1e3 * sd.stats["kernel_ms"] / (K4 * 6), 1e3 * sd.adjoint_stats["kernel_ms"] / (K4 * 6)
    out["c4_8_designs"] = {"lattice": "64x64-cell kagome + contact, 8 designs, forward + design gradient through the problem layer", "steps": K4,
                           "fwd_stage_us": f_us, "adj_stage_us": a_us, "value": K4 * n4 * 8 / wall4,
                           "device_value": K4 * n4 * 8 / (1e-3 * (sd.stats["kernel_ms"] + sd.adjoint_stats["kernel_ms"])),
                           "fwd_frac": BYTES_FWD_STAGE_KAGOME * n4 * 8 / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "adj_frac": BYTES_ADJ_STAGE_KAGOME * n4 * 8 / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "checkpoint": {1: "records", 2: "segments"}.get(sd.adjoint_stats.get("checkpoint_records", 0)) or "stages/state",
                           "kernels": {"forward": BUILD_NAMES.get(int(sd.stats.get("tile_kernels", 0))),
                                       "adjoint": BUILD_NAMES.get(int(sd.adjoint_stats.get("tile_kernels", 0)))}}
    sd.engine.close()


def c3_as_written_leg(args, device, sync, steps=50000):
    """C3 as BASELINE.json / SURVEY 8(d) write it, whole: 128 x 128 quads with contact, **50 000 steps** over 2/f, forward + adjoint
    w.r.t. the design, pulse delayed by 0.1/f, 2x2 target shifted by (N//6, N//5) = (21, 25), all members.  4.4 TB of stage records
    do not fit, so the solve runs at the SEGMENTS level (the reverse sweep re-runs one output interval at a time: 3 s launches per
    step instead of 2 s) -- what a user of C3 gets, next to the headline's K-step window at the records level.  ~18 s.
    (`--as-written-steps K` shortens it: the window then starts with the pulse, t0 = 0.1/f; below ~4 000 steps the wave has not
    reached the far target and objective and gradient are numerical dust.)"""
    t_d = 0.1 / FREQ
    keep = os.environ.get("DFX_CHECKPOINT")
    os.environ["DFX_CHECKPOINT"] = "segments"
    os.environ["DFX_STREAMS"] = str(args.streams)
    try:
        members = min(args.members, args.as_written_members)     # (16 designs: 32 measured the same rate -- 8.22e8 vs 8.17e8 -- in twice the time)
        fw, obj, designs = c3_problem(args.size, 3, members, device=device, input_delay=t_d,
                                      target_shift=(args.size // 6, args.size // 5))
        eng = fw.solve_dynamics.engine
        eng.reserve(steps, steps // SPI + 2, keep_trajectory=True)
        t_start = 0.0 if steps >= 50000 else t_d
        prepare(fw, designs, 2 * SPI, t_start=t_d)
        execute(fw, obj)                                         # warm-up: two output intervals, same kernels
        prepare(fw, designs, steps, t_start=t_start)
        spin_up(fw)
        sync()
        t0 = time.perf_counter()
        res = execute(fw, obj)
        sync()
        wall = time.perf_counter() - t0
        gnorm = grad_norm(res)
        eng.close()
    finally:
        if keep is None:
            os.environ.pop("DFX_CHECKPOINT", None)
        else:
            os.environ["DFX_CHECKPOINT"] = keep
    n_units = args.size * args.size
    streams = res["streams"]
    # per reverse step and stream: 6 re-run forward stages + 6 reverse stages
    f_us = 1e3 * res["fwd_ms"] / max(1.0, res["fwd_launches"] / streams)
    n_adj = res["adj_launches"] / 2.0 / streams
    a_us = max(1e-9, (1e3 * res["adj_ms"] - n_adj * f_us) / n_adj)
    per_step_bytes = 6 * BYTES_FWD_STAGE + 48 + 6 * BYTES_ADJ_STAGE          # what ONE forward + ONE reverse pass need
    total = steps * n_units * members
    return {"value": total / wall, "unit": "timesteps*units/s", "steps": steps, "window": "the whole horizon: steps 0..50000, t = 0 .. 2/f" if steps >= 50000 else f"steps 2500..{2500 + steps} of 50000 (t0 = 0.1/f: the pulse starts)",
            "members_per_gpu": members, "checkpoint": res.get("checkpoint"), "input_delay_s": t_d,
            "target_shift": [args.size // 6, args.size // 5], "target_blocks": [int(b) for b in obj.target_blocks],
            "device_ms": {"forward": res["fwd_ms"], "adjoint": res["adj_ms"], "wall": 1e3 * wall},
            "adjoint_over_forward": res["adj_ms"] / res["fwd_ms"], "launches": {"forward": res["fwd_launches"], "adjoint": res["adj_launches"]},
            "objective": [float(x) for x in np.atleast_1d(res["objective"])][:4], "grad_norm": gnorm,
            "end_to_end_frac_of_hbm_peak": per_step_bytes * total / wall / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "kernel": "k_adj_stage<nonlinear,contact>", "regime": f"{streams} member groups on concurrent streams",
                         "launch_period_us": a_us, "achieved": BYTES_ADJ_STAGE * n_units * members / (a_us * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": BYTES_ADJ_STAGE * n_units * members / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "forward_launch_period_us": f_us,
                         "forward_frac": BYTES_FWD_STAGE * n_units * members / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS}}


C4_DESIGNS = 64                 # BASELINE config 4: 64 designs in all, seeds 100 .. 163


def c4_problem(members, steps, device=0, lib=None):
    """BASELINE config 4 (SURVEY 8(d) "C4"): 64x64-cell kagome (8 192 triangles), notebook constants of
    kagome_focusing_3dp_pla_shims.ipynb cell 7, contact + damping, pulse on the left edge, 3/f horizon, target kinetic energy,
    through the caller the reference uses (problems/kagome_focusing.py restated in difflexmm_amd/problems.py)."""
    from difflexmm_amd.problems import KagomeFocusingForward, TargetKineticEnergy
    n1 = n2 = 64
    rho, ksh, kr, cell = 6.18e-9, 1.19, 1.5, 20.0
    damping = 0.0186 * np.array([2 * math.sqrt(0.070175913225 * rho * cell ** 2 * ksh)] * 2 +
                                [2 * math.sqrt(0.0009477510275 * rho * cell ** 4 * kr)]) * np.ones((2 * n1 * n2, 1))
    n_out = 40
    spi = max(1, steps // n_out)
    fw = KagomeFocusingForward(n1_cells=n1, n2_cells=n2, cell_size=cell, bond_length=2.25, k_stretch=120.0, k_shear=ksh, k_rot=kr,
                               density=rho, damping=damping, amplitude=0.5 * cell, loading_rate=FREQ, input_delay=0.1 / FREQ,
                               n_excited_blocks=2, simulation_time=spi * n_out * DT,      # the full config: 75 000 steps of DT = 3/f
                               n_timepoints=n_out + 1, use_contact=True, k_contact=kr, min_angle=-15 * math.pi / 180,
                               cutoff_angle=-10 * math.pi / 180, steps_per_interval=spi, batch=members, device=device, _lib=lib)
    obj = TargetKineticEnergy(fw, (2, 2), (n1 // 6, n2 // 5))
    return fw, obj, spi * n_out


def run_c4(args, comm, comm_info, world, rank, local_rank):
    """`--workload c4 --gpus N`: the 64 designs are dealt to the ranks in contiguous equal chunks (strong scaling: the total is
    fixed), every rank evaluates objective + design gradient of its chunk as ONE batch through the problem layer (host-side design
    maps included, as a user of problems/kagome_focusing.py pays them), objectives are combined with ONE all-gather."""
    from difflexmm_amd import _binding as B
    from difflexmm_amd import ensemble
    if C4_DESIGNS % world:
        sys.exit(f"bench.py --workload c4: {C4_DESIGNS} designs do not split evenly over {world} ranks")
    members = C4_DESIGNS // world
    K = max(40, args.steps - args.steps % 40)
    lo, hi = ensemble.shard_bounds(C4_DESIGNS, rank, world)
    # designs per engine call: all of this rank's.  (The full 75 000-step horizon with 64 designs on ONE GPU: the stage records of one
    # output interval are 385 GB -- the engine re-runs such an interval in pieces from restart states its forward pass leaves behind, at no
    # extra cost: 7.65e8 in one call, the same as two calls of 32.  Should an engine still refuse a width, half as many designs per call,
    # the calls one after the other: TargetKineticEnergy.value_and_grad.)
    per_call = members
    while True:
        fw, obj, K = c4_problem(per_call, K, device=local_rank)
        designs = []
        for seed in range(100 + lo, 100 + hi):
            rng = np.random.default_rng(seed)
            designs.append(tuple(rng.uniform(-0.3, 0.3, sh) for sh in fw.geometry.design_shapes()))
        try:
            obj.value_and_grad(designs[:per_call])      # allocations, graphs, clocks (and: does this width fit?)
            break
        except RuntimeError as e:
            if "cannot allocate" not in str(e) or per_call % 2:
                raise
            fw.solve_dynamics.engine.close()
            per_call //= 2
    B.device_synchronize(local_rank); comm.barrier(); B.device_synchronize(local_rank)
    obj.device_ms_forward = obj.device_ms_adjoint = 0.0
    t0 = time.perf_counter()
    vals, grads = obj.value_and_grad(designs)
    B.device_synchronize(local_rank); comm.barrier(); B.device_synchronize(local_rank)
    wall = float(comm.all_reduce([time.perf_counter() - t0], "max")[0])
    allv = ensemble.gather_objectives(vals, C4_DESIGNS, comm)
    sd = fw.solve_dynamics
    dev_ms = comm.all_reduce([obj.device_ms_forward, obj.device_ms_adjoint], "max")
    if rank == 0:
        n_units = fw.geometry.n_blocks
        gn = float(np.sqrt(sum(float(np.vdot(a, a)) for g in grads for a in g)))
        line = {"metric": "timesteps*rigid-units/s (forward + design gradient)", "value": K * n_units * C4_DESIGNS / wall,
                "unit": "timesteps*units/s", "n_gpus": world, "steps": K, "warmup": 1, "ms_per_step": 1e3 * wall / K,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"C4: 64x64-cell kagome ({n_units} units), nonlinear ligaments + damping + angle contact, pulse drive, "
                                       f"fixed-step Dopri5, {K} steps over {fw.simulation_time:.3e} s, {C4_DESIGNS} designs in all "
                                       f"(seeds 100..163), forward + gradient w.r.t. the three shift fields through KagomeFocusingForward / "
                                       "TargetKineticEnergy (host-side design maps inside the timed region)",
                           "designs_total": C4_DESIGNS, "members_per_gpu": members, "designs_per_engine_call": per_call,
                           "collective": comm_info["collective"],
                           "ranks_seen": comm_info.get("ranks_seen"), "rccl_version": comm_info.get("rccl_runtime"),
                           "checkpoint": {1: "records", 2: "segments"}.get(sd.adjoint_stats.get("checkpoint_records", 0))
                           or ("stages" if sd.adjoint_stats.get("stage_checkpoint") else "state"),
                           "concurrent_streams": int(sd.adjoint_stats.get("streams", 1))},
                "device_ms": {"forward": float(dev_ms[0]), "adjoint": float(dev_ms[1]), "wall": 1e3 * wall},
                "device_only_value": K * n_units * C4_DESIGNS / (1e-3 * float(dev_ms[0] + dev_ms[1])),
                "objective": [float(x) for x in allv[:8]], "objectives_gathered": int(len(allv)), "grad_norm_rank0": gn}
        # the dominant kernel: one reverse stage of this rank's designs (a launch, or a stage of the persistent loop); device time of the
        # sweep / stages.  At the segments level the reverse sweep also re-runs the forward pass: its share is taken out first.
        f_us = 1e3 * float(dev_ms[0]) / (K * 6)
        a_ms = float(dev_ms[1]) - (float(dev_ms[0]) if sd.adjoint_stats.get("checkpoint_records", 0) == 2 else 0.0)
        a_us = max(1e-9, 1e3 * a_ms / (K * 6))
        builds = {"forward": BUILD_NAMES.get(int(sd.stats.get("tile_kernels", 0))), "adjoint": BUILD_NAMES.get(int(sd.adjoint_stats.get("tile_kernels", 0)))}
        a_us, f_us = a_us * per_call / members, f_us * per_call / members       # (sequential calls: stages of ONE call)
        line["roofline"] = stage_roofline("reverse stage <nonlinear,contact>, 3-node blocks", BYTES_ADJ_STAGE_KAGOME, n_units * per_call, a_us,
                                          kernels=builds, members_per_stage=per_call,
                                          measured_with="HIP events around the reverse sweep / (steps x 6 stages), max over ranks")
        line["roofline_forward_kernel"] = stage_roofline("forward stage <nonlinear,contact>, 3-node blocks", BYTES_FWD_STAGE_KAGOME, n_units * per_call, f_us)
        line["csrc"] = source_ids()
        if world == 1 and not args.no_launch_bound:
            # what ONE rank of an 8-GPU run integrates: 8 of the 64 designs, same steps, same call (round-4 verdict #3).  The ratio
            # predicts the strong scaling of the data path before any collective or imbalance; it is NOT a measured 8-GPU number.
            fw8, obj8, _ = c4_problem(C4_DESIGNS // 8, K, device=local_rank)
            d8 = designs[:C4_DESIGNS // 8]
            obj8.value_and_grad(d8)
            B.device_synchronize(local_rank)
            t8 = time.perf_counter()
            obj8.value_and_grad(d8)
            B.device_synchronize(local_rank)
            w8 = time.perf_counter() - t8
            s8 = fw8.solve_dynamics
            v8 = K * n_units * (C4_DESIGNS // 8) / w8
            line["per_rank_of_8"] = {"designs": C4_DESIGNS // 8, "value": v8, "wall_ms": 1e3 * w8,
                                     "device_ms": {"forward": s8.stats["kernel_ms"], "adjoint": s8.adjoint_stats["kernel_ms"]},
                                     "device_only_value": K * n_units * (C4_DESIGNS // 8) / (1e-3 * (s8.stats["kernel_ms"] + s8.adjoint_stats["kernel_ms"])),
                                     "kernels": {"forward": BUILD_NAMES.get(int(s8.stats.get("tile_kernels", 0))),
                                                 "adjoint": BUILD_NAMES.get(int(s8.adjoint_stats.get("tile_kernels", 0)))}}
            line["predicted_strong_scaling"] = {"at_8_gpus": 8.0 * v8 / line["value"],
                                                "how": "8 x value(8 designs on this GPU) / value(64 designs on this GPU): two 1-GPU runs of the same call; "
                                                       "no collective, no imbalance, not a measured multi-GPU number"}
            s8.engine.close()
        if world == 1 and not args.no_cpu_baseline:
            def make(lib):
                fwc, objc, _ = c4_problem(1, 40, lib=lib)
                rng = np.random.default_rng(100)
                return fwc, objc, [tuple(rng.uniform(-0.3, 0.3, sh) for sh in fwc.geometry.design_shapes())]
            line["cpu_baseline"] = cpu_baseline(64, 100, n_steps=40, repeats=3, budget_s=30.0, make=make,
                                                what="the same 64x64-cell kagome lattice (8192 units)")
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()
    sd.engine.close()


def run_c2(args, comm, comm_info, world, rank, local_rank):
    """`--workload c2`: BASELINE configs[1] -- 32x32 quads, nonlinear ligaments + damping, 10 000 fixed Dopri5 steps over 2/f, forward
    only, ONE member per GPU (what the config says; 64 waves in all: the launch-bound end of the path), and the same lattice
    `--c2-members` wide beside it.  `--steps K` times exactly K steps (default of this workload: the whole 10 000)."""
    from difflexmm_amd import _binding as B
    K = 10000 if args.steps == 5000 else max(1, args.steps)
    W = max(0, args.warmup)

    def sync():
        B.device_synchronize(local_rank)

    legs = {}
    for name, members in (("one_member", 1), ("batched", max(1, args.c2_members))):
        fw, obj, designs = c2_problem(members, device=local_rank)
        for _ in range(1 if W else 0):
            c2_solve(fw, designs, min(K, max(W, SPI)), sync)
        sync(); comm.barrier(); sync()
        wall, st = c2_solve(fw, designs, K, sync)
        comm.barrier()
        wall = float(comm.all_reduce([wall], "max")[0])
        us = 1e3 * st["kernel_ms"] / (K * 6)
        legs[name] = {"members_per_gpu": members, "value": K * 1024 * members * world / wall, "wall_ms": 1e3 * wall, "device_ms": st["kernel_ms"],
                      "device_value": K * 1024 * members * world / (st["kernel_ms"] * 1e-3), "launches": st["launches"],
                      "roofline": stage_roofline("forward stage <nonlinear, no contact>", BYTES_FWD_STAGE_C2, 1024 * members, us,
                                                 kernels=BUILD_NAMES.get(int(st.get("tile_kernels", 0))), members_per_stage=members,
                                                 measured_with="HIP events around the forward pass / (steps x 6 stages)")}
        fw.solve_dynamics.engine.close()
    if rank == 0:
        one = legs["one_member"]
        bat = legs["batched"]
        # the line leads with BOTH widths (round-5 verdict): the config as written is ONE 32x32 member -- 64 waves on a 1 024-SIMD chip, bound by the
        # hand-off latency of the stage loop, not by bytes -- and the same lattice batched is what the engine is designed for
        line = {"metric": "timesteps*rigid-units/s (forward)", "value": one["value"], "value_batched": bat["value"],
                "roofline_frac_one_member": one["roofline"]["frac"], "roofline_frac_batched": bat["roofline"]["frac"],
                "unit": "timesteps*units/s", "n_gpus": world, "steps": K,
                "warmup": W, "ms_per_step": one["wall_ms"] / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"C2: 32x32 quads (1024 units), nonlinear ligaments + damping, no contact, pulse drive, fixed-step Dopri5 "
                                       f"dt={C2_DT:.3e}s, {K} of 10000 steps, forward only; `value`: 1 member per GPU as the config is written "
                                       f"(launch-latency bound: 64 waves), `value_batched`: {bat['members_per_gpu']} members side by side",
                           "members_per_gpu": 1, "members_per_gpu_batched": bat["members_per_gpu"], "collective": comm_info["collective"],
                           "ranks_seen": comm_info.get("ranks_seen")},
                "roofline": one["roofline"], "one_member": one, "batched": legs["batched"], "csrc": source_ids()}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(32, 2, n_steps=2000, repeats=3, budget_s=30.0, make=lambda lib: c2_problem(1, lib=lib), adjoint=False,
                                                what="the same 32x32 lattice")
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()


PAPER_UNITS = 24 * 16


def paper_problem(members, device=0, lib=None, angle_deg=35.0):
    """The reference's OWN call (round-5 verdict #1): the lattice, constants and tolerances of the notebooks (24 x 16 quads, spacing 15 mm,
    hinge 0.15 spacing, k = 120 / 1.19 / 1.5, contact -15 / -10 deg, pulse 0.5 spacing at 30 Hz delayed by 0.1 / f, 200 outputs over 2 / f,
    `odeint(rtol=1e-8, atol=1e-4)` -- problems/quads_focusing.py:73-74 with the notebook's atol; tests/notebook_kat.py pins this very
    problem to two numbers the reference printed), objective = kinetic energy of a 2 x 2 target (problems/quads_focusing.py:432-471),
    value + gradient w.r.t. the design.  Member 0 is the notebook's initial design (rotated squares at 35 degrees), the others perturb it."""
    from difflexmm_amd.problems import QuadsFocusingForward, TargetKineticEnergy
    n1, n2, spacing = 24, 16, 15.0
    rho, ks, ksh, kr = 6.18e-9, 120.0, 1.19, 1.5
    damping = 0.0186 * np.array([2 * math.sqrt(0.36125 * rho * spacing ** 2 * ksh)] * 2 +
                                [2 * math.sqrt(0.02175026 * rho * spacing ** 4 * kr)]) * np.ones((n1 * n2, 1))
    fw = QuadsFocusingForward(n1_blocks=n1, n2_blocks=n2, spacing=spacing, bond_length=0.15 * spacing, k_stretch=ks, k_shear=ksh, k_rot=kr,
                              density=rho, damping=damping, use_contact=True, k_contact=kr, min_angle=-15 * math.pi / 180,
                              cutoff_angle=-10 * math.pi / 180, amplitude=0.5 * spacing, loading_rate=FREQ, input_delay=0.1 / FREQ,
                              n_excited_blocks=2, loaded_side="left", input_shift=0, simulation_time=2 / FREQ, n_timepoints=200,
                              atol=1e-4, rtol=1e-8, batch=members, device=device, _lib=lib)
    obj = TargetKineticEnergy(fw, (2, 2), (5, 3))
    base = fw.geometry.get_design_from_rotated_square(angle_deg * math.pi / 180)
    designs = [tuple(np.array(b) for b in base)]
    for m in range(1, members):
        rng = np.random.default_rng(1000 + m)
        designs.append(tuple(b + rng.uniform(-0.02 * spacing, 0.02 * spacing, b.shape) for b in base))
    return fw, obj, designs


def paper_eval(fw, obj):
    """objective + raw gradient of the designs `prepare` left on the device: what `jit(value_and_grad(objective))` is in the reference."""
    sd = fw.solve_dynamics
    sd.solve_resident(fw.state0, fw.timepoints, keep_trajectory=True, want_fields=False)
    vals, raw = sd.kinetic_energy_value_and_raw(obj.target_blocks)
    st, ast, ad = sd.stats, sd.adjoint_stats, getattr(sd, "adaptive_stats", None)
    two_pass = st.get("step_control") == "adaptive-grid"
    counts = sd.engine.adaptive_step_counts() if hasattr(sd.engine, "adaptive_step_counts") else None
    info = {"forward_passes": 2 if two_pass else 1,
            "attempts_max": int(((ad if two_pass else st)["rhs_evals"] - 2) // 6),
            "accepted_per_member": [int(x) for x in counts.sum(1)] if counts is not None else None,
            "reverse_steps": int(ast["steps"]),
            "device_ms": {"adaptive_forward": float((ad if two_pass else st)["kernel_ms"]), "frozen_grid_forward": float(st["kernel_ms"]) if two_pass else 0.0,
                          "reverse": float(ast["kernel_ms"])},
            "launches": {"adaptive_forward": int((ad if two_pass else st)["launches"]), "frozen_grid_forward": int(st["launches"]) if two_pass else 0,
                         "reverse": int(ast["launches"])},
            "kernels": {"adaptive_forward": (ad if two_pass else st).get("tile_kernels", 0), "reverse": ast.get("tile_kernels", 0)}}
    return np.asarray(vals, dtype=float), raw, info


def run_paper(args, comm, comm_info, world, rank, local_rank):
    """`--workload paper`: one "step" of this workload is ONE whole evaluation (objective + gradient) of the paper's problem with the
    reference's own integrator settings; `--steps K` evaluations are timed after `--warmup W` (defaults 5 / 1).  Two widths: ONE design
    (the reference's call) and `--paper-members` designs side by side (weak scaling over ranks: every rank its own designs)."""
    from difflexmm_amd import _binding as B
    K = 5 if args.steps == 5000 else max(1, args.steps)
    W = 1 if args.warmup == 250 else max(0, args.warmup)

    def sync():
        B.device_synchronize(local_rank)

    legs = {}
    for name, members in (("one_design", 1), ("batched", max(1, args.paper_members))):
        fw, obj, designs = paper_problem(members, device=local_rank)
        sd = fw.solve_dynamics
        t0 = time.perf_counter()
        sd.prepare([fw.control_params(d) for d in designs])
        prep_ms = 1e3 * (time.perf_counter() - t0)
        for _ in range(W):
            paper_eval(fw, obj)
        sync(); comm.barrier(); sync()
        t0 = time.perf_counter()
        for _ in range(K):
            vals, raw, info = paper_eval(fw, obj)
        sync(); comm.barrier(); sync()
        wall = float(comm.all_reduce([time.perf_counter() - t0], "max")[0]) / K
        acc = info["accepted_per_member"] or [info["reverse_steps"]] * members
        unit_steps = PAPER_UNITS * float(sum(acc))
        dms = info["device_ms"]
        stage_us_fwd = 1e3 * dms["adaptive_forward"] / max(1, info["attempts_max"] * 6)
        stage_us_rev = 1e3 * dms["reverse"] / max(1, info["reverse_steps"] * 6)
        legs[name] = dict(info, members_per_gpu=members, value=unit_steps * world / wall, wall_ms_per_evaluation=1e3 * wall, host_prepare_ms=prep_ms,
                          device_only_value=unit_steps * world / (1e-3 * sum(dms.values())),
                          objective=[float(x) for x in vals[:4]],
                          grad_norm=float(np.sqrt(sum(float(np.vdot(a, a)) for a in raw.values()))),
                          us_per_stage={"adaptive_forward_attempt": stage_us_fwd, "reverse": stage_us_rev},
                          roofline=stage_roofline("reverse stage <nonlinear,contact>", BYTES_ADJ_STAGE, PAPER_UNITS * members, stage_us_rev,
                                                  kernels=BUILD_NAMES.get(int(info["kernels"]["reverse"])), members_per_stage=members,
                                                  measured_with="HIP events around the reverse sweep / (steps x 6 stages)"),
                          roofline_forward=stage_roofline("adaptive forward stage <nonlinear,contact>", BYTES_FWD_STAGE, PAPER_UNITS * members,
                                                          stage_us_fwd, kernels=BUILD_NAMES.get(int(info["kernels"]["adaptive_forward"])),
                                                          measured_with="HIP events around the adaptive pass / (attempts x 6 evaluations): controller, "
                                                                        "error norm and dense output included"))
        sd.engine.close()
    if rank == 0:
        one = legs["one_design"]
        line = {"metric": "timesteps*rigid-units/s (adaptive forward + gradient, accepted steps)", "value": one["value"], "unit": "timesteps*units/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": one["wall_ms_per_evaluation"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "paper: 24x16 quads (384 units), nonlinear ligaments + damping + angle contact, delayed pulse, 200 outputs over 2/f, "
                                       "adaptive Dopri5 rtol=1e-8 atol=1e-4 (jax.experimental.ode semantics), objective + design-reaching gradient; "
                                       "one step of this workload = one whole evaluation; value counts ACCEPTED steps x units",
                           "members_per_gpu": 1, "collective": comm_info["collective"], "ranks_seen": comm_info.get("ranks_seen")},
                "roofline": one["roofline"], "one_design": one, "batched": legs["batched"], "csrc": source_ids()}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = paper_cpu_baseline()
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()


def paper_cpu_baseline(repeats=5):
    """The C++ port of the oracle on the same evaluation, one design: adaptive pass + reverse sweep (on the frozen grid: the port has no
    dense-output adjoint), all usable cores and one thread, median of `repeats` after one warm-up."""
    import ctypes
    from oracle.cpu import load, load_native
    lib, build_flags = load_native()
    if lib is None:
        lib, build_flags = load(), build_flags + " (oracle/cpu/Makefile)"
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    fw, obj, designs = paper_problem(1, lib=lib)
    fw.solve_dynamics.prepare([fw.control_params(d) for d in designs])
    ncpu = usable_cpus()
    legs = {}
    for nt in ([1, ncpu] if gomp is not None and ncpu > 1 else [1]):
        if gomp is not None:
            gomp.omp_set_num_threads(nt)
        paper_eval(fw, obj)
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            _, _, info = paper_eval(fw, obj)
            times.append(time.perf_counter() - t0)
        acc = info["accepted_per_member"] or [info["reverse_steps"]]
        legs[nt] = PAPER_UNITS * float(sum(acc)) / float(np.median(times))
    best = max(legs, key=legs.get)
    return {"value": legs[best], "unit": "timesteps*units/s", "cores": best, "kind": "port", "one_thread": legs.get(1),
            "all_cores": legs.get(ncpu) if ncpu > 1 else None, "usable_cpus": ncpu, "build": build_flags,
            "sample": f"the whole evaluation (adaptive forward, {info['forward_passes']} forward pass(es), reverse sweep) of one design of the same "
                      f"24x16 lattice; C++ port of the oracle, 1 thread and {ncpu} threads, 1 warm-up + median of {repeats} runs each"}


def usable_cpus():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host is
    often granted a handful; spinning 256 OpenMP threads on them takes minutes per step)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = float(txt[0])
                period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(size, seed, n_steps=100, repeats=5, budget_s=40.0, make=None, adjoint=True, what=None):
    """The CPU port of the oracle (same algorithm, same tableau, OpenMP over blocks) on a bounded sample of the same
    workload: `n_steps` Dopri5 steps forward + adjoint of the same lattice, one member.  SURVEY 8(d) protocol: 1 thread and
    all usable cores, one warm-up run, median of `repeats` timed runs each; `value` is the faster of the two.  A 4-step probe
    per leg shortens the sample when the host is too slow for the time budget (and says so).
    `make(lib) -> (forward problem, objective, designs)` selects another lattice (default: C3), `adjoint=False` a forward-only sample."""
    import ctypes
    from oracle.cpu import load, load_native
    lib, build_flags = load_native()              # -march=native, compiled on the host that is timed
    if lib is None:
        lib, build_flags = load(), build_flags + " (oracle/cpu/Makefile)"
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    fw, obj, designs = make(lib) if make else c3_problem(size, seed, 1, lib=lib)
    n_units = fw.geometry.n_blocks
    ncpu = usable_cpus()
    thread_counts = [1, ncpu] if gomp is not None and ncpu > 1 else [1]
    legs, notes = {}, []
    for nt in thread_counts:
        if gomp is not None:
            gomp.omp_set_num_threads(nt)
        prepare(fw, designs, 4, spi=2)
        execute(fw, obj, adjoint, spi=2)                                # touch everything once
        prepare(fw, designs, 4, spi=2)
        t0 = time.perf_counter()
        execute(fw, obj, adjoint, spi=2)
        per_step = (time.perf_counter() - t0) / 4
        n = n_steps
        leg_budget = budget_s / len(thread_counts)
        if per_step * n * (repeats + 1) > leg_budget:
            n = int(max(4, leg_budget / (per_step * (repeats + 1)) // 2 * 2))
            notes.append(f"{nt} thread(s): {n} steps instead of {n_steps} (time budget)")
        spi = min(n, 50)
        prepare(fw, designs, n, spi=spi)
        execute(fw, obj, adjoint, spi=spi)                              # warm-up (first call excluded, scripts/pulse_RS.py:93-108)
        times = []
        for _ in range(repeats):
            prepare(fw, designs, n, spi=spi)                  # same split as the GPU leg: inputs prepared outside the timed region
            t0 = time.perf_counter()
            execute(fw, obj, adjoint, spi=spi)
            times.append(time.perf_counter() - t0)
        legs[nt] = n * n_units / float(np.median(times))
    best = max(legs, key=legs.get)
    return {"value": legs[best], "unit": "timesteps*units/s", "cores": best, "kind": "port",
            "one_thread": legs.get(1), "all_cores": legs.get(ncpu) if ncpu > 1 else None, "usable_cpus": ncpu,
            "logical_cpus": os.cpu_count(), "build": build_flags,
            "sample": f"{n_steps} Dopri5 steps {'forward+adjoint' if adjoint else 'forward only'} of {what or f'the same {size}x{size} lattice'}, 1 member; C++ port of the "
                      f"oracle (OpenMP over blocks), 1 thread and {ncpu} threads, 1 warm-up + median of {repeats} runs each"
                      + ("; " + "; ".join(notes) if notes else "")}


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: start the N rank processes (fresh interpreters, before this process has made any GPU
    call), relay rank 0's output, return the worst exit code."""
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    uid_file = os.path.join(tempfile.gettempdir(), f"dfx_uid_{port}_{os.getpid()}")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DFX_UID_FILE=uid_file, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    return max(p.wait() for p in procs)


def make_comm(args, world, rank, local_rank):
    """Communicator of the N > 1 run + what it really is.  RCCL (inside libdfx) is the collective; a TCP control channel comes first:
    the unique id travels over it and the ranks agree on go / no-go BEFORE anyone enters ncclCommInitRank.  `--backend rccl` (the
    default) is strict: if RCCL does not come up on N distinct devices every rank exits non-zero -- no silent TCP stand-in;
    `--backend socket` is the explicit rehearsal (several ranks on one GPU)."""
    from difflexmm_amd import ensemble
    if world == 1:
        return ensemble.SerialComm(), {"collective": "none (1 rank)", "ranks_seen": 1}
    ctrl = ensemble.SocketComm(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                               int(os.environ.get("DFX_SOCKET_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 11)))
    if args.backend == "socket":
        return ctrl, {"collective": "socket (rehearsal)", "ranks_seen": ctrl.world}
    err, comm = "", None
    try:
        comm = ensemble.init_from_env("rccl", device=local_rank, ctrl=ctrl)
    except Exception as e:            # noqa: BLE001 -- reported, not swallowed
        err = f"{type(e).__name__}: {e}"
    ok = ctrl.all_reduce([0.0 if err else 1.0], "min")[0] > 0
    if not ok:
        if rank == 0 or err:
            print(f"bench: rank {rank}: RCCL did not come up ({err or 'on another rank'}); --backend rccl does not fall back", file=sys.stderr)
        ctrl.close()
        sys.exit(3)
    info = dict(comm.info(), collective="rccl")
    ctrl.close()
    return comm, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--members", type=int, default=0,
                    help="independent designs per GPU integrated side by side (grid.y).  0 (default) = choose: 32 while the stage records of "
                         "the run (432 B x units x steps each) fit the free HBM -- a launch over 32 designs spreads its ramp and tail (~3.3 us) "
                         "over twice the work of one over 16: reverse launch 0.74 instead of 0.70 of the roofline, the same job rate --, else "
                         "16 (longer runs: the richest checkpoint level that fits, as before)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="member groups advanced concurrently, one HIP stream each")
    ap.add_argument("--no-single", action="store_true", help="skip the extra 1-member reference measurement")
    ap.add_argument("--outputs", default="hbm", choices=("hbm", "host"),
                    help="where the timed job leaves its gradients: hbm (device pointers; the download is timed after the region and "
                         "reported) | host (pinned views: the D2H copy is inside the timed region)")
    ap.add_argument("--no-roofline-leg", action="store_true", help="skip the separate 1-stream per-launch measurement")
    ap.add_argument("--no-launch-bound", action="store_true", help="skip the launch-bound regimes' block (C2 x 1, C3 x 1, C4 x 8 designs)")
    ap.add_argument("--c2-members", type=int, default=64, help="--workload c2: width of the second, batched leg")
    ap.add_argument("--backend", default="rccl", help="collective of the N>1 run: rccl (inside libdfx; strict: exits non-zero if it "
                                                      "does not come up on N distinct GPUs) | socket (rehearsal on one GPU)")
    ap.add_argument("--paper-members", type=int, default=32, help="--workload paper: width of the second, batched leg")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5", "paper"],
                    help="c3: 128x128 quads, fixed designs per GPU (weak scaling; the headline).  c2: BASELINE config 2 -- 32x32 quads, "
                         "nonlinear ligaments + damping, 10 000 fixed steps, forward only, one member (and --c2-members beside it).  c4: BASELINE config 4 -- 64 kagome "
                         "designs (64x64 cells) in all, sharded over the ranks, forward + design gradient through the problem layer, "
                         "one all-gather of objectives (strong scaling).  c5: BASELINE config 5 -- the multi-input inverse design as "
                         "an ensemble of --c5-members designs in lock-step, --c5-iterations objective evaluations each "
                         "(examples/multi_input_ensemble.py; strong scaling)")
    ap.add_argument("--c5-members", type=int, default=256)
    ap.add_argument("--c5-iterations", type=int, default=4)
    ap.add_argument("--no-as-written", action="store_true", help="skip the extra C3-as-written leg (segments checkpoint, paper's pulse "
                                                                 "delay and target placement, 2500 steps)")
    ap.add_argument("--as-written-members", type=int, default=16, help="designs per GPU of the C3-as-written leg (at most the job's own width)")
    ap.add_argument("--as-written-steps", type=int, default=50000, help="steps of the C3-as-written leg (default: the whole 50 000-step horizon, ~18 s)")
    ap.add_argument("--all-ranks-device", type=int, default=-1, help="rehearsal only: put every rank on this device")
    ap.add_argument("--input-delay", type=float, default=0.0, help="pulse delay in s (C3 text: 0.1/f = 3.33e-3)")
    ap.add_argument("--contact-cutoff-deg", type=float, default=-10.0,
                    help="void angle below which the contact penalty engages (C3: -10, never reached; 60 with --contact-min-deg 30 "
                         "engages every narrow void of the 25-degree design: contact branches and their gradient traffic everywhere)")
    ap.add_argument("--contact-min-deg", type=float, default=-15.0, help="void angle at which the penalty diverges (C3: -15)")
    ap.add_argument("--target-shift", type=int, nargs=2, default=None, help="target placement (C3 text: 21 25); default: next to the drive")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not os.path.exists(os.path.join(ROOT, "difflexmm_amd", "libdfx.so")):
        # build artefact missing (fresh checkout): compile it the way __graft_entry__.build() does, before any GPU call
        if rank == 0:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "difflexmm_amd", "csrc")], stdout=subprocess.DEVNULL)
        else:
            while not os.path.exists(os.path.join(ROOT, "difflexmm_amd", "libdfx.so")):
                time.sleep(0.5)
    if args.workload == "c5":
        # the example forks its host workers before anything touches the GPU and brings up its own communicator
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import multi_input_ensemble
        return multi_input_ensemble.main(["--members", str(args.c5_members), "--iterations", str(args.c5_iterations), "--backend", args.backend,
                                          "--all-ranks-device", str(args.all_ranks_device), "--json"])
    from difflexmm_amd import _binding as B
    if args.all_ranks_device >= 0:
        local_rank = args.all_ranks_device
        # several PROCESSES on one GPU: the persistent stage loop needs all workgroups of a launch resident, and the account that keeps
        # concurrent persistent launches within the chip is per process -- the rehearsal keeps one launch per stage
        os.environ["DFX_PERSIST"] = "0"
    comm, comm_info = make_comm(args, world, rank, local_rank)
    collective = comm_info["collective"]
    if args.workload == "c4":
        return run_c4(args, comm, comm_info, world, rank, local_rank)
    if args.workload == "c2":
        return run_c2(args, comm, comm_info, world, rank, local_rank)
    if args.workload == "paper":
        return run_paper(args, comm, comm_info, world, rank, local_rank)
    os.environ["DFX_STREAMS"] = str(args.streams)
    K = max(1, args.steps)                      # EXACTLY K steps are timed
    W = max(0, args.warmup)
    adjoint = not args.forward_only
    requested_members = args.members if args.members > 0 else "auto"
    if args.members <= 0:
        free_b, total_b = B.mem_info(local_rank)
        fits32 = 432.0 * (max(K, W, ROOFLINE_LEG_STEPS) + 1) * args.size * args.size * 32 < free_b - 0.05 * total_b - 2e9
        args.members = int(comm.all_reduce([32.0 if fits32 else 16.0], "min")[0])                 # the same width on every rank
    prob = dict(input_delay=args.input_delay, target_shift=args.target_shift, contact_cutoff_deg=args.contact_cutoff_deg,
                contact_min_deg=args.contact_min_deg)
    reserve_steps = max(K, W, ROOFLINE_LEG_STEPS)
    if adjoint:
        # What the forward pass keeps for the reverse sweep (the engine takes the richest level that fits; decide here, by the same
        # rule, so that every leg below runs the same kernels): records 432 B per unit and step (reverse launches read their stage
        # record directly), stages 72 + 120 B (records rebuilt elementwise), state 72 B (records recomputed), segments: nothing but
        # the outputs -- the reverse sweep re-runs one output interval at a time; memory independent of the horizon, so the full
        # 50 000 steps keep all 16 members.
        free_b, total_b = B.mem_info(local_rank)
        if "DFX_CHECKPOINT" not in os.environ and "DFX_STAGE_CHECKPOINT" not in os.environ:
            n_ck = max(K, W, ROOFLINE_LEG_STEPS)
            units = args.size * args.size * args.members
            room = free_b - 0.05 * total_b - 2e9
            level = ("records" if 432.0 * (n_ck + 1) * units < room else
                     "stages" if (72.0 * (n_ck + 1) + 120.0 * n_ck) * units < room else
                     "state" if 72.0 * (n_ck + 1) * units < room else "segments")
            # where the persistent stage loop serves the job (its waves fit on the chip in at most two launches: 3 reverse workgroups of
            # 4 waves on each of 256 CUs) the engine goes from records straight to segments (engine_forward.hip, choose_checkpoint)
            if level != "records" and args.members * ((args.size * args.size + 15) // 16) <= 2 * 3 * 256 * 4:
                level = "segments"
            level = ["records", "stages", "state", "segments"][int(comm.all_reduce(
                [float(["records", "stages", "state", "segments"].index(level))], "max")[0])]      # same kernels on every rank
            os.environ["DFX_CHECKPOINT"] = level
        if args.members < args.streams:
            args.streams = args.members
            os.environ["DFX_STREAMS"] = str(args.streams)

    def sync():
        B.device_synchronize(local_rank)

    # (1) per-launch roofline of the two stage kernels: ONE stream, every launch integrates all `members` designs, a fixed
    #     ROOFLINE_LEG_STEPS steps whatever K is (a 20-step region is too short to average over).  This is the regime
    #     rocprofv3 can observe (its kernel trace serialises queues): `python bench.py --streams 1 --steps 250` under
    #     rocprofv3 --kernel-trace --stats reports the same average durations (profiles/).  Rank 0, before the timed job, on its
    #     own engine (closed again before the big checkpoint of the timed job is allocated).
    rr = None
    if rank == 0 and not args.no_roofline_leg:
        os.environ["DFX_STREAMS"] = "1"
        timed_level = os.environ.get("DFX_CHECKPOINT")
        if timed_level == "segments":          # the segments level runs the records kernels inside every output interval
            os.environ["DFX_CHECKPOINT"] = "records"
        fwr, objr, desr = c3_problem(args.size, 3 + 1000 * rank, args.members, device=local_rank, **prob)
        os.environ["DFX_STREAMS"] = str(args.streams)
        Kr = ROOFLINE_LEG_STEPS
        fwr.solve_dynamics.engine.reserve(Kr, Kr // SPI + 2, keep_trajectory=adjoint)
        run_once(fwr, objr, desr, Kr, adjoint=adjoint)           # warm-up of the same length (graphs, clocks)
        sync()
        rr = run_once(fwr, objr, desr, Kr, adjoint=adjoint)
        rr.pop("grads", None)
        fwr.solve_dynamics.engine.close()
        del fwr, objr
        if timed_level is not None:
            os.environ["DFX_CHECKPOINT"] = timed_level
    single = None
    if rank == 0 and world == 1 and args.members > 1 and adjoint and not args.no_single:
        # the same config with ONE design per GPU (launch-bound: one wave per SIMD), for reference.  The engine chooses its checkpoint
        # level itself here (the level pinned above is the 16-member job's: a 5 000-step default run pins "stages", at which one member's
        # reverse sweep would keep the stage launches although its records fit)
        pinned_level = os.environ.pop("DFX_CHECKPOINT", None)
        fw1, obj1, des1 = c3_problem(args.size, 3, 1, device=local_rank, **prob)
        K1 = min(max(K, 250), 2500)
        fw1.solve_dynamics.engine.reserve(K1, K1 // SPI + 2, keep_trajectory=True)
        run_once(fw1, obj1, des1, K1)
        prepare(fw1, des1, K1)
        spin_up(fw1)
        sync()
        t1 = time.perf_counter()
        r1 = execute(fw1, obj1)
        sync()
        w1 = time.perf_counter() - t1
        # device time per Runge-Kutta stage (a launch each, or one stage of the persistent loop): the sweep's HIP events / stages
        f1_us, a1_us = 1e3 * r1["fwd_ms"] / (6.0 * K1), 1e3 * r1["adj_ms"] / (6.0 * K1)
        n1u = args.size * args.size
        single = {"members_per_gpu": 1, "steps": K1, "value": K1 * args.size * args.size / w1,
                  "forward_only_value": K1 * args.size * args.size / (r1["fwd_ms"] * 1e-3),
                  "fwd_stage_us": f1_us, "adj_stage_us": a1_us,
                  "kernels": {"forward": BUILD_NAMES.get(r1.get("fwd_build")), "adjoint": BUILD_NAMES.get(r1.get("adj_build"))},
                  "launches": {"forward": r1["fwd_launches"], "adjoint": r1["adj_launches"]},
                  "roofline": {"bound": "hbm", "kernel": "reverse stage <nonlinear,contact>", "stage_us": a1_us,
                               "achieved": BYTES_ADJ_STAGE * n1u / (a1_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": BYTES_ADJ_STAGE * n1u / (a1_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "forward_frac": BYTES_FWD_STAGE * n1u / (f1_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                  "checkpoint": r1.get("checkpoint"),
                  "device_ms": {"forward": r1["fwd_ms"], "adjoint": r1["adj_ms"]}}
        fw1.solve_dynamics.engine.close()
        del fw1, obj1, r1
        if pinned_level is not None:
            os.environ["DFX_CHECKPOINT"] = pinned_level
    fw, obj, designs = c3_problem(args.size, 3 + 1000 * rank, args.members, device=local_rank, **prob)
    device_outputs = adjoint and args.outputs == "hbm" and hasattr(fw.solve_dynamics.engine.lib, "dfx_kinetic_value_and_grad_device")
    fw.solve_dynamics.engine.reserve(max(K, W), max(K, W) // SPI + 2, keep_trajectory=adjoint)
    if W:
        run_once(fw, obj, designs, W, adjoint=adjoint, device_outputs=device_outputs)
    # hipGraphs are instantiated on first use: make sure every segment length the K timed steps replay has been used once
    # (a full output interval and the shorter last interval), whatever W was
    if K >= SPI and (W < SPI or W % SPI):
        run_once(fw, obj, designs, SPI, adjoint=adjoint, device_outputs=device_outputs)
    if K % SPI:
        run_once(fw, obj, designs, K % SPI, adjoint=adjoint, device_outputs=device_outputs)

    def barrier():
        sync()
        comm.barrier()
        sync()

    tp0 = time.perf_counter()
    prepare(fw, designs, K)                    # inputs resident in HBM before the timed region
    host_prepare_ms = 1e3 * (time.perf_counter() - tp0)   # design -> ControlParams -> packed arrays -> H2D, reported, not timed
    spin_up(fw)
    barrier()
    t0 = time.perf_counter()
    res = execute(fw, obj, adjoint=adjoint, device_outputs=device_outputs)
    barrier()
    wall = time.perf_counter() - t0
    out_bytes = sum(getattr(a, "nbytes", 0) for a in (res.get("grads") or {}).values())
    fetch_s = fetch(res)                       # outputs over PCIe to the host: after the timed region, reported
    gnorm = grad_norm(res)
    wall_host_outputs = None
    if device_outputs and K <= 2500:
        # the PCIe-inclusive rate of the same job: one more region through the host-output call (gradients arrive as pinned views)
        run_once(fw, obj, designs, K, adjoint=adjoint)
        prepare(fw, designs, K)
        spin_up(fw)
        barrier()
        th = time.perf_counter()
        execute(fw, obj, adjoint=adjoint)
        barrier()
        wall_host_outputs = float(comm.all_reduce([time.perf_counter() - th], "max")[0])
    objective = res["objective"] if res["objective"] is not None else np.zeros(args.members)
    wall = float(comm.all_reduce([wall], "max")[0])                    # MAX over ranks
    objective = comm.all_gather(objective).ravel()                     # the single collective of the path: objectives over xGMI
    if rank == 0:
        n_units = args.size * args.size
        total_units_steps = K * n_units * args.members * world
        streams = res["streams"]

        def per_launch(r, n_streams):
            """(fwd launch us, adj launch us): region device time (HIP events on the engine's stream) / launches per stream."""
            f_us = 1e3 * r["fwd_ms"] / max(1.0, r["fwd_launches"] / n_streams)
            a_us = None
            if r["adj_launches"] and r.get("checkpoint") in ("records", "stages"):
                a_us = 1e3 * r["adj_ms"] / (r["adj_launches"] / n_streams)     # reverse stages only (no recompute launches)
            elif r["adj_launches"] and r.get("checkpoint") == "segments":
                n_adj = r["adj_launches"] / 2.0 / n_streams           # per reverse step: 6 re-run forward stages + 6 reverse stages
                a_us = max(1e-9, (1e3 * r["adj_ms"] - n_adj * f_us) / n_adj)
            elif r["adj_launches"]:
                n_adj = r["adj_launches"] * 6.0 / 11.0 / n_streams   # per reverse step: 5 recomputed forward stages + 6 reverse stages
                a_us = max(1e-9, (1e3 * r["adj_ms"] - n_adj * (5.0 / 6.0) * f_us) / n_adj)
            return f_us, a_us

        fw.solve_dynamics.engine.close()
        leg = rr if rr is not None else res
        leg_streams = 1 if rr is not None else streams
        fwd_us, adj_us = per_launch(leg, leg_streams)
        mpl = args.members / leg_streams
        traffic = load_pmc_traffic()

        def roof(kernel, bytes_per_unit, us, extra=None):
            ach = bytes_per_unit * n_units * mpl / (us * 1e-6) / 1e9
            d = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                 "traffic": None, "algorithmic_bytes_per_unit": bytes_per_unit, "bytes_per_launch": bytes_per_unit * n_units * mpl,
                 "launch_us": us, "members_per_launch": mpl,
                 "measured_with": (f"separate leg: 1 stream, {ROOFLINE_LEG_STEPS} steps, HIP events on the engine's stream around the "
                                   "region / launches" if rr is not None else "the timed region / launches per stream")}
            d.update(extra or {})
            return d

        line = {
            "metric": "timesteps*rigid-units/s (forward + adjoint)" if adjoint else "timesteps*rigid-units/s (forward)",
            "value": total_units_steps / wall, "unit": "timesteps*units/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * wall / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3: {args.size}x{args.size} quads, nonlinear ligaments + damping + angle contact, "
                                   f"pulse drive, fixed-step Dopri5 dt={DT:.3e}s, {K} of 50000 steps, "
                                   f"{'forward + adjoint: ControlParams-shaped gradient per design (node vectors, undeformed void angles, inertia: 245760 values; the 66048 design shifts follow by a linear host-side map outside the solver boundary)' if adjoint else 'forward only'}",
                       "members_per_gpu": args.members, "members_requested": requested_members, "concurrent_streams": streams,
                       "checkpoint": res.get("checkpoint"), "integrator": "dopri5-fixed",
                       "steps_per_output": SPI, "input_delay_s": args.input_delay, "contact_deg": [args.contact_min_deg, args.contact_cutoff_deg],
                       "target_blocks": [int(b) for b in obj.target_blocks], "collective": collective,
                       "ranks_seen": comm_info.get("ranks_seen"), "rccl_version": comm_info.get("rccl_runtime")},
            "forward_only_value": K * n_units * args.members * world / (res["fwd_ms"] * 1e-3),
            "device_ms": {"forward": res["fwd_ms"], "adjoint": res["adj_ms"], "wall": 1e3 * wall},
            "host_prepare_ms": host_prepare_ms, "value_with_host_prepare": total_units_steps / (wall + 1e-3 * host_prepare_ms),
            # the boundary's PCIe legs, never part of `value`: inputs in (host_prepare_ms: design -> packed arrays -> H2D) and
            # gradients out (the accumulators stay in HBM inside the timed region; `--outputs host` times the pinned-copy call instead)
            "outputs": {"where": "hbm" if device_outputs else "host (pinned views, inside the timed region)",
                        "library_calls": ("dfx_forward_kinetic_value_and_grad (one call)" if res.get("fused_call") else
                                          "dfx_forward_grid + dfx_kinetic_value_and_grad" + ("_device" if device_outputs else "")),
                        "gradient_bytes": int(out_bytes), "download_ms": 1e3 * fetch_s,
                        "value_with_outputs_on_host": total_units_steps / (wall_host_outputs or (wall + fetch_s)),
                        "measured_with": ("a second region of the same K steps through dfx_kinetic_value_and_grad (pinned views)"
                                          if wall_host_outputs else "timed region + download")},
            "launches": {"forward": res["fwd_launches"], "adjoint": res["adj_launches"]},
            # RHS evaluations per second (whole-lattice evaluations of every member; the reverse sweep adds one
            # Hessian-vector product per forward evaluation, counted as one more each)
            "rhs_evals_per_s": (12 if adjoint else 6) * K * args.members * world / wall,
            "objective": [float(x) for x in np.atleast_1d(objective)][:8], "grad_norm": gnorm,
        }
        if adj_us:
            # the dominant kernel: the reverse stage.  SURVEY 8(d) counts 624 B per unit; with the stage checkpoint 5 of the 6
            # reverse launches of a step also rebuild a stage record (+60 B on average: the forward's stage combine moved here)
            line["roofline"] = roof("k_adj_stage<nonlinear,contact>", BYTES_ADJ_STAGE, adj_us,
                                    {"checkpoint": leg.get("checkpoint"),
                                     "own_count_bytes_per_unit": BYTES_ADJ_STAGE + (60 if leg.get("checkpoint") == "stages" else 0),
                                     "traffic": None if traffic is None else traffic.get("k_adj_stage_bytes_per_member_launch", 0) * mpl or None,
                                     "traffic_source": None if traffic is None else traffic.get("source")})
            line["roofline_forward_kernel"] = roof("k_fwd_stage<nonlinear,contact>", BYTES_FWD_STAGE, fwd_us,
                                                   {"traffic": None if traffic is None else traffic.get("k_fwd_stage_bytes_per_member_launch", 0) * mpl or None})
        else:
            line["roofline"] = roof("k_fwd_stage<nonlinear,contact>", BYTES_FWD_STAGE, fwd_us,
                                    {"traffic": None if traffic is None else traffic.get("k_fwd_stage_bytes_per_member_launch", 0) * mpl or None,
                                     "traffic_source": None if traffic is None else traffic.get("source")})
        # SURVEY 8(d), caveat H5: the fp64 VALU rate beside the byte rate (instruction counts per launch from the committed SQ
        # counter passes, 64 lanes per wave instruction, FMA = 2 flops; launch time measured in this run)
        if traffic is not None and "counters" in traffic:
            def valu(which, us):
                c = traffic["counters"].get(which, {})
                if not c or not us:
                    return None
                flop = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0)
                               + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0))
                flop *= mpl / float(traffic.get("members", mpl))
                return {"fp64_flop_per_launch": flop, "achieved_tflops": flop / (us * 1e-6) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                        "frac": flop / (us * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS}
            line["fp64_valu"] = {"k_fwd_stage": valu("fwd", fwd_us), "k_adj_stage": valu("adj", adj_us)}
        if streams > 1:
            # (2) the timed job itself: `streams` member groups overlap on the chip; aggregate algorithmic bytes / region time
            f_eff, a_eff = per_launch(res, streams)
            agg = {"concurrent_streams": streams, "members_per_launch": args.members / streams,
                   "fwd_stage_period_us": f_eff, "fwd_achieved": BYTES_FWD_STAGE * n_units * args.members / (f_eff * 1e-6) / 1e9}
            agg["fwd_frac"] = agg["fwd_achieved"] / HBM_PEAK_GBS
            if a_eff:
                agg["adj_stage_period_us"] = a_eff
                agg["adj_achieved"] = BYTES_ADJ_STAGE * n_units * args.members / (a_eff * 1e-6) / 1e9
                agg["adj_frac"] = agg["adj_achieved"] / HBM_PEAK_GBS
            line["roofline_concurrent"] = agg
        # end to end in SURVEY 8(d)'s per-step accounting: 2 064 B forward + 3 792 B reverse per step and unit
        per_step_bytes = (6 * BYTES_FWD_STAGE) + (48 + 6 * BYTES_ADJ_STAGE if adjoint else 0)
        line["end_to_end_frac_of_hbm_peak"] = per_step_bytes * total_units_steps / wall / 1e9 / HBM_PEAK_GBS
        if single is not None:
            line["single_system"] = single
        if world == 1 and adjoint and not args.no_launch_bound:
            line["launch_bound"] = launch_bound_block(local_rank, sync, single)
        line["csrc"] = source_ids()
        if world == 1 and adjoint and not args.no_as_written and args.size == 128 and args.as_written_steps > 0:
            line["c3_as_written"] = c3_as_written_leg(args, local_rank, sync, steps=args.as_written_steps)
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.size, 3)
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()


KERNEL_SOURCES = ("dfx_physics.h", "dfx_plan.h", "dfx_stage.h", "dfx_kernels.h", "stage_builds_adj.hip")   # what the two stage kernels are compiled from
ENGINE_SOURCES = KERNEL_SOURCES + ("dfx_persist.h", "dfx_persist_api.h", "dfx_engine.h", "engine_launch.hip", "engine_forward.hip", "engine_adaptive.hip",
                                  "engine_reverse.hip", "engine_abi.hip", "dfx_persist.hip", "dfx_comm.hip", "Makefile")


def csrc_digest(files=KERNEL_SOURCES):
    """sha256 over an EXPLICIT list of engine sources (default: the headers the stage kernels are compiled from -- what a counter
    file of those kernels belongs to), by name and content: identifies the build on the GPU box, where the snapshot has no .git.
    (Round-4 advice: a glob also hashed whatever else sat in the directory.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "difflexmm_amd", "csrc")
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def source_ids():
    """What every bench line says about the tree it ran on: digest of the stage kernels' sources, digest of all engine sources, and the
    commit when the tree has a .git (the GPU box's snapshot has none)."""
    head = None
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:       # noqa: BLE001
        pass
    return {"stage_kernel_sources_sha256": csrc_digest(), "engine_sources_sha256": csrc_digest(ENGINE_SOURCES), "git_head": head}


def load_pmc_traffic():
    """HBM bytes per member and launch of the two stage kernels from this round's committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json: FETCH_SIZE with the guide's x2 correction for gfx950 + WRITE_SIZE, separate passes), or None.
    Counters cannot be read from inside the run; the file names the command they were collected with and the sources they were collected
    for (`csrc_sha256`, `commit`): counters of another engine build are NOT reported -- `traffic` stays null and `traffic_source` says why."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(p))
    except Exception:       # noqa: BLE001
        return None
    now = csrc_digest()
    if t.get("csrc_sha256") != now:
        return {"source": f"STALE, not reported: profiles/pmc_traffic.json was collected for engine sources {t.get('csrc_sha256')} "
                          f"(commit {t.get('commit')}), this tree's difflexmm_amd/csrc is {now}"}
    t["source"] = f"{t.get('source')}; engine sources {now} = commit {t.get('commit')}"
    return t


if __name__ == "__main__":
    main()
