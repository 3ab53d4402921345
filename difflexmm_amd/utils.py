"""Parameter trees of the solver API -- field-for-field the reference's ``difflexmm/utils.py:9-163``
(NumPy arrays / floats instead of JAX arrays) -- plus the pickle IO of ``utils.py:166-201``."""
import pickle
from pathlib import Path
from typing import Any, Dict, NamedTuple, Optional, Union

import numpy as np


class SolutionData(NamedTuple):
    """utils.py:9-25."""
    block_centroids: Any
    centroid_node_vectors: Any
    bond_connectivity: Any
    timepoints: Any
    fields: Any


class EigenmodeData(NamedTuple):
    """utils.py:28-45: what ``linear_mode_analysis`` results are stored as (eigenvalues (n_modes,), fields (n_modes, 2, n_blocks, 3))."""
    block_centroids: Any
    centroid_node_vectors: Any
    eigenvalues: Any
    fields: Any


SolutionType = Union[SolutionData, EigenmodeData]      # utils.py:46


class GeometricalParams(NamedTuple):
    """utils.py:48-59."""
    block_centroids: Any
    centroid_node_vectors: Any


class LigamentParams(NamedTuple):
    """utils.py:62-77; stiffnesses are scalars or (n_bonds,) arrays."""
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    reference_vector: Any


class StretchingTorsionalSpringParams(NamedTuple):
    """utils.py:80-91: zero-length springs (``stretching_torsional_spring_energy``); scalars or (n_bonds,) arrays."""
    k_stretch: Any
    k_rot: Any


BondParams = Union[LigamentParams, StretchingTorsionalSpringParams]


class ContactParams(NamedTuple):
    """utils.py:97-111."""
    min_angle: Any
    cutoff_angle: Any
    k_contact: Any


class MagneticParams(NamedTuple):
    """utils.py:114-125 (the reference defines the container; no energy of its ``energy.py`` reads it)."""
    dipole_angles: Any
    dipole_strengths: Any


class MechanicalParams(NamedTuple):
    """utils.py:128-142."""
    bond_params: BondParams
    density: Any
    inertia: Optional[Any] = None
    damping: Any = 0.
    contact_params: Optional[ContactParams] = None


class ControlParams(NamedTuple):
    """utils.py:145-163."""
    geometrical_params: GeometricalParams
    mechanical_params: MechanicalParams
    magnetic_params: Optional[MagneticParams] = None
    loading_params: Dict = dict()
    constraint_params: Dict = dict()


def save_data(path_or_filename: Union[str, Path], data: object):
    """utils.py:166-180."""
    path = Path(path_or_filename)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as file:
        pickle.dump(data, file)
        print("Data saved at " + str(path))


def load_data(path_or_filename: Union[str, Path]):
    """utils.py:183-201."""
    with open(path_or_filename, "rb") as file:
        data = pickle.load(file)
    if isinstance(data, (SolutionData, EigenmodeData)):
        return type(data)(*(np.asarray(a) if isinstance(a, np.ndarray) else a for a in data))
    return data


def is_scalar(x):
    """utils.py:204-213."""
    return np.asarray(x).shape == ()
