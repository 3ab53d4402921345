"""Rigid-body kinematics on the host -- the names of ``difflexmm/kinematics.py`` for post-processing code (the time loop evaluates the
same maps inside the stage kernels, ``csrc/dfx_physics.h``)."""
import numpy as np

from .energy import _block_to_node_displacement, block_to_node_kinematics  # noqa: F401
from .geometry import DOFsInfo


def build_constrained_kinematics(geometry, constrained_block_DOF_pairs, constrained_DOFs_fn=lambda t, **kwargs: 0):
    """kinematics.py:40-81: returns ``constrained_kinematics(free_DOFs, t, constraint_params={}) -> (n_blocks, 3)``: zeros, the
    prescribed values ``constrained_DOFs_fn(t, **constraint_params)`` on the constrained DOFs, the free DOFs where they belong."""
    free_ids, con_ids, all_ids = DOFsInfo(geometry.n_blocks, constrained_block_DOF_pairs)

    def constrained_kinematics(free_DOFs, t, constraint_params=dict()):
        all_DOFs = np.zeros(len(all_ids))
        if len(con_ids) != 0:
            all_DOFs[con_ids] = constrained_DOFs_fn(t, **constraint_params)
        all_DOFs[free_ids] = free_DOFs
        return all_DOFs.reshape(geometry.n_blocks, 3)

    return constrained_kinematics
