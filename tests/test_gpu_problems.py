"""-m gpu: the problem layer end to end on the HIP engine (BASELINE config C5 in miniature: quads_focusing_multi_input on
the paper's 24x16 lattice, three inputs, a short optimisation loop) and the batched ensemble evaluation."""
import math

import numpy as np
import pytest

from difflexmm_amd import ensemble
from difflexmm_amd import problems as P

pytestmark = pytest.mark.gpu


def _fw(side, shift, batch=1, lib=None):
    fw = P.QuadsFocusingForward(
        n1_blocks=24, n2_blocks=16, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5,
        density=6.18e-9, damping=0.0186 * np.array([2 * math.sqrt(0.36125 * 6.18e-9 * 225 * 1.19)] * 2 +
                                                   [2 * math.sqrt(0.02175026 * 6.18e-9 * 15.0 ** 4 * 1.5)]) * np.ones((384, 1)),
        amplitude=7.5, loading_rate=300.0, input_delay=1e-4, n_excited_blocks=2, loaded_side=side, input_shift=shift,
        simulation_time=8e-3, n_timepoints=21, use_contact=True, k_contact=1.5, min_angle=-15 * math.pi / 180,
        cutoff_angle=-10 * math.pi / 180, steps_per_interval=40, batch=batch, _lib=lib)
    fw.setup()
    return fw


def _design(fw, seed):
    rng = np.random.default_rng(seed)
    base = fw.geometry.get_design_from_rotated_square(25 * math.pi / 180)
    return tuple(b + rng.uniform(-0.3, 0.3, b.shape) for b in base)


def test_multi_input_optimisation_loop_24x16(hip_lib, cpu_lib):
    fws = [_fw("left", 0), _fw("right", -2), _fw("bottom", -4)]
    mi = P.MultiInputTargetKineticEnergy(fws, (2, 2), (4, 3), weights=(1.0, 1.0, 1.0))
    x0 = _design(fws[0], 1000)
    # one evaluation against the CPU port of the oracle
    ref = P.MultiInputTargetKineticEnergy([_fw("left", 0, lib=cpu_lib), _fw("right", -2, lib=cpu_lib), _fw("bottom", -4, lib=cpu_lib)],
                                          (2, 2), (4, 3), weights=(1.0, 1.0, 1.0))
    v, g = mi.value_and_grad(x0)
    vr, gr = ref.value_and_grad(x0)
    assert v > 0 and abs(v - vr) / vr < 1e-9
    for a, b in zip(g, gr):
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-7
    opt = P.OptimizationProblem(mi)
    opt.run_optimization(x0, 2, lower_bound=-3.0, upper_bound=3.0, min_void_angle=5 * math.pi / 180,
                         min_block_angle=5 * math.pi / 180, min_edge_length=1.0, verbose=False)
    assert opt.objective_values[-1] > opt.objective_values[0]


def test_batched_ensemble_equals_one_by_one(hip_lib):
    """8 designs integrated side by side (one solver with batch=8, 4 concurrent streams) == the same designs one by one."""
    fwb = _fw("left", 0, batch=8)
    fw1 = _fw("left", 0)
    designs = [_design(fw1, 1000 + i) for i in range(8)]
    vals_b, grads_b, _ = ensemble.evaluate_ensemble(P.TargetKineticEnergy(fwb, (2, 2), (4, 3)), designs)
    obj1 = P.TargetKineticEnergy(fw1, (2, 2), (4, 3))
    for i, d in enumerate(designs):
        v, g = obj1.value_and_grad(d)
        assert abs(vals_b[i] - v) <= 1e-12 * abs(v)
        for a, b in zip(grads_b[i], g):
            assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max()


def test_native_rccl_communicator_single_rank(hip_lib, tmp_path):
    """The collective entry points of libdfx (RCCL inside the library: dfx_comm_init / dfx_gather_objectives /
    dfx_reduce_grads) on a one-rank communicator -- what one GPU can check: unique id hand-over through the file, communicator
    creation on the device, an all-gather and the three all-reduces, the device helpers."""
    from difflexmm_amd import _binding as B
    comm = ensemble.RcclComm(0, 1, 0, str(tmp_path / "uid"))
    assert hip_lib.dfx_comm_size(comm._c) == 1 and hip_lib.dfx_comm_rank(comm._c) == 0
    x = np.arange(5.0) + 0.25
    assert np.array_equal(comm.all_gather(x), x[None])
    for op in ("sum", "max", "min"):
        assert np.array_equal(comm.all_reduce(x.reshape(5, 1), op), x.reshape(5, 1))
    comm.barrier()
    assert np.array_equal(ensemble.gather_objectives([1.0, 2.0], 2, comm), [1.0, 2.0])
    comm.close()
    free_b, total_b = B.mem_info(0)
    assert 0 < free_b <= total_b and total_b > 200e9
    B.device_synchronize(0)


@pytest.mark.parametrize("batch", [1, 3])
def test_response_data_reduced_on_the_device(hip_lib, batch):
    """per-ligament strain-energy and per-block kinetic-energy histories from the device-resident fields
    (k_response), against the oracle's strains and the host formulas."""
    from .test_problems import check_response_data
    check_response_data(None, batch)


def test_inputs_evaluated_in_turn_share_one_checkpoint(hip_lib):
    """Engines of a multi-input objective whose inputs run one after the other keep their trajectory checkpoints in ONE set of
    buffers (dfx_share_checkpoint): same objective and gradient as engines with checkpoints of their own, and a reverse sweep on a
    handle whose checkpoint another handle has overwritten meanwhile is refused."""
    def build(streams):
        fws = []
        for side, shift in (("left", 0), ("right", -2)):
            fw = P.QuadsFocusingForward(
                n1_blocks=24, n2_blocks=16, spacing=15.0, bond_length=2.25, k_stretch=120.0, k_shear=1.19, k_rot=1.5, density=6.18e-9,
                damping=1e-4 * np.ones((384, 3)), amplitude=7.5, loading_rate=300.0, input_delay=1e-4, n_excited_blocks=2,
                loaded_side=side, input_shift=shift, simulation_time=4e-3, n_timepoints=11, use_contact=True, k_contact=1.5,
                min_angle=-15 * math.pi / 180, cutoff_angle=-10 * math.pi / 180, steps_per_interval=40, batch=2, streams=streams)
            fw.setup()
            fws.append(fw)
        return P.MultiInputTargetKineticEnergy(fws, (2, 2), (4, 3), weights=(1.0, 0.5)), fws
    own, _ = build(1)                         # one stream each: concurrent inputs, every engine its own checkpoint
    shared, fws = build(2)                    # two member groups each: inputs in turn, ONE checkpoint
    assert own.concurrent_inputs and not shared.concurrent_inputs
    designs = [_design(fws[0], 1000), _design(fws[0], 1001)]
    v0, g0 = own.value_and_grad(designs)
    v1, g1 = shared.value_and_grad(designs)
    assert np.allclose(v0, v1, rtol=1e-12)
    for a, b in zip(g0, g1):
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-10 * np.abs(y).max()
    # input 0's checkpoint has been overwritten by input 1's forward pass: its reverse sweep must refuse to run
    with pytest.raises(RuntimeError, match="overwritten by a solve of another handle"):
        fws[0].solve_dynamics.kinetic_energy_value_and_raw(shared.target_blocks)
