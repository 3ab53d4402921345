"""difflexmm_amd -- MI355X-native engine for DifFlexMM's hot path (force assembly + explicit RK integrator +
discrete adjoint), behind the reference's ``setup_dynamic_solver`` / ``ControlParams`` API.

Host code is NumPy + ctypes; the time loop runs in hand-written HIP kernels (``csrc/``, built into
``libdfx.so``).  There is no CPU fallback: importing is cheap, but creating a solver without the built
library raises ``RuntimeError``.
"""
__version__ = "0.1.0"

from .dynamics import linear_mode_analysis, setup_dynamic_solver  # noqa: F401
from .utils import (ContactParams, ControlParams, EigenmodeData, GeometricalParams, LigamentParams,  # noqa: F401
                    MagneticParams, MechanicalParams, SolutionData, StretchingTorsionalSpringParams)
