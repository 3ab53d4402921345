"""Hinge characterisation: fitting the ligament stiffnesses to measured force-displacement curves
(``problems/hinge_characterization.py``).

A rotating-squares sample (``HingeForward``) or a quad sample with given shifts (``HingeQuadsForward``) is pulled, pushed or sheared in
displacement control between its clamped bottom and top rows; the design variables are ``(k_stretch, k_shear, k_rot)``; the response is
the reaction force on the driven DOFs of the top row -- the elastic force ``dE/du`` there -- against the applied displacement, and the
objective of ``HingeResponseError`` is the mean squared distance to the measured curves (hinge_characterization.py:621-648).

The reference differentiates all of it with ``jax.grad``.  Here the dynamics and its reverse sweep run on the engine, and so does the
reaction force: a second, unconstrained handle with one member per output time evaluates ``a = -M^-1 dE/du`` for all output
configurations in one call (``dfx_rhs``), and its Hessian-vector hook (``dfx_rhs_vjp``) gives, again in one call, both pieces of the
gradient that do not come from the reverse sweep -- ``K(u_t) e_R`` (the cotangent of the displacement history, which the reverse sweep
then carries back to the stiffnesses) and the explicit derivative of the elastic force w.r.t. the stiffnesses."""
from dataclasses import InitVar, dataclass
from typing import Any, Dict, List, Optional

import numpy as np

from . import energy as E
from . import loading as L
from .dynamics import setup_dynamic_solver
from .geometry import QuadGeometry, RotatedSquareGeometry
from .utils import ContactParams, ControlParams, GeometricalParams, LigamentParams, MechanicalParams, SolutionData


def hinge_constraints(geometry, loading_type):
    """hinge_characterization.py:95-130: top and bottom rows clamped in all three DOFs, the top row driven along y (tension +1,
    compression -1) or x (shear +1).  Returns (pairs, loading vector, reaction pairs)."""
    n1, nb = geometry.n1_blocks, geometry.n_blocks
    blocks = np.concatenate([np.arange(nb - n1, nb), np.arange(n1)])              # top row, bottom row
    pairs = np.stack([np.concatenate([blocks] * 3), np.repeat([0, 1, 2], len(blocks))], 1).astype(np.int64)
    if loading_type not in ("tension", "compression", "shear"):
        raise ValueError("Loading type should be either tension, compression, or shear!")
    dof = 0 if loading_type == "shear" else 1
    top_row = np.where(pairs[:, 1] == dof)[0][:n1]
    vec = np.zeros(len(pairs))
    vec[top_row] = -1.0 if loading_type == "compression" else 1.0
    return pairs, vec, pairs[top_row]


def resample(x, y, n_timepoints):
    """hinge_characterization.py:546-551."""
    return np.interp(np.linspace(np.min(x), np.max(x), n_timepoints), x, y)


def enforce_bounds(x, lower=None, upper=None):
    """hinge_characterization.py:554-560."""
    x = np.asarray(x, dtype=float)
    if lower is not None:
        x = np.where(x <= lower, lower, x)
    if upper is not None:
        x = np.where(x >= upper, upper, x)
    return x


class _HingeBase:
    """What the two samples share: setup, solve, force-displacement and its derivatives."""

    def setup(self):
        g, design = self._make_geometry()
        self.geometry = g
        self._block_centroids, self._centroid_node_vectors = g.block_centroids(*design), g.centroid_node_vectors(*design)
        self.bond_connectivity, self.reference_bond_vectors = g.bond_connectivity(), g.reference_bond_vectors()
        k_ref, mass_ref = self.k_stretch, self.density * g.spacing ** 2                       # :85-93
        damping_ref = np.array([(k_ref * mass_ref) ** 0.5, (k_ref * mass_ref) ** 0.5, (k_ref * mass_ref) ** 0.5 * g.spacing ** 2])
        self.damping_values = self.damping * damping_ref * np.ones((g.n_blocks, 3))
        pairs, vec, self.reaction_block_DOF_pairs = hinge_constraints(g, self.loading_type)
        self.constrained_block_DOF_pairs = pairs
        strain = E.build_strain_energy(self.bond_connectivity, E.ligament_energy_linearized if self.linearized_strains else E.ligament_energy)
        energy = E.combine_block_energies(strain, E.build_contact_energy(self.bond_connectivity)) if self.use_contact else strain
        self.potential_energy = energy
        # ramp up to the target displacement (:134-139): amplitude * min(t * loading_rate, 1) on the driven DOFs
        self.solve_dynamics = setup_dynamic_solver(
            g, energy, constrained_block_DOF_pairs=pairs, constrained_DOFs_fn=L.Ramp(vec, amplitude="amplitude", rate="loading_rate"),
            damped_blocks=np.arange(g.n_blocks), rtol=self.rtol, atol=self.atol, steps_per_interval=self.steps_per_interval,
            device=self.device, _lib=self._lib)
        # reaction forces: the elastic force of EVERY block DOF at the output configurations -- an unconstrained handle, one member
        # per output time
        self._force = setup_dynamic_solver(g, energy, batch=self.n_timepoints, device=self.device, _lib=self._lib)
        self.timepoints = np.linspace(0, 1.0 / self.loading_rate, self.n_timepoints)          # :167-168
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.is_setup = True

    def applied_displacement(self, t, amplitude, loading_rate):
        t = np.asarray(t, dtype=float)
        return amplitude * np.where(t < 1.0 / loading_rate, t * loading_rate, 1.0)

    def control_params(self, k_values):
        k_stretch, k_shear, k_rot = k_values
        return ControlParams(
            geometrical_params=GeometricalParams(self._block_centroids, self._centroid_node_vectors),
            mechanical_params=MechanicalParams(
                bond_params=LigamentParams(k_stretch, k_shear, k_rot, self.reference_bond_vectors), density=self.density,
                damping=self.damping_values,
                contact_params=ContactParams(min_angle=self.min_angle, cutoff_angle=self.cutoff_angle, k_contact=self.k_contact)),
            constraint_params=dict(amplitude=self.amplitude, loading_rate=self.loading_rate))

    def solve(self, k_values, keep_trajectory=False):
        """(SolutionData, ControlParams) as the reference's ``forward`` (:170-215)."""
        cp = self.control_params(k_values)
        fields = self.solve_dynamics(self.state0, self.timepoints, cp, keep_trajectory=keep_trajectory)
        self.solution_data = SolutionData(self._block_centroids, self._centroid_node_vectors, self.bond_connectivity, self.timepoints, fields)
        return self.solution_data, cp

    def _load_force_engine(self, control_params):
        flat = self._force._flatten(control_params)
        T = self.n_timepoints
        self._force.engine.set_params(**{k: np.broadcast_to(v, (T,) + np.shape(v)).copy() for k, v in flat.items()})
        return np.asarray(flat["inertia"], dtype=float).reshape(-1, 3)

    def force_displacement(self, solution_data, control_params):
        """[applied displacement, reaction force * force_multiplier] (:225-244): sum over the driven top-row DOFs of the elastic force
        dE/du at every output configuration."""
        inertia = self._load_force_engine(control_params)
        T, n = self.n_timepoints, self.geometry.n_blocks
        y = np.zeros((T, 2, n, 3))
        y[:, 0] = solution_data.fields[:, 0]
        acc = self._force.engine.rhs(y, 0.0)[:, 1]                        # -dE/du / m (no velocity, no loading)
        rb, rd = self.reaction_block_DOF_pairs[:, 0], self.reaction_block_DOF_pairs[:, 1]
        force_history = -(acc[:, rb, rd] * inertia[rb, rd]).sum(1)
        applied_u = self.applied_displacement(solution_data.timepoints, **control_params.constraint_params)
        return np.array([applied_u, force_history * self.force_multiplier])

    def force_vjp(self, solution_data, control_params, force_bar):
        """Cotangent ``force_bar`` (T,) of the (multiplied) force history -> (cotangent of the fields, explicit d/d(k_stretch, k_shear,
        k_rot)): one batched Hessian-vector call.  With w = -force_bar * mult * m on the reaction DOFs as cotangent of the accelerations,
        (da/du)^T w = K e_R force_bar mult and d(w.a)/dk = force_bar mult d(dE/du)_R/dk."""
        inertia = self._load_force_engine(control_params)
        T, n = self.n_timepoints, self.geometry.n_blocks
        y = np.zeros((T, 2, n, 3))
        y[:, 0] = solution_data.fields[:, 0]
        lam = np.zeros((T, 2, n, 3))
        rb, rd = self.reaction_block_DOF_pairs[:, 0], self.reaction_block_DOF_pairs[:, 1]
        lam[:, 1, rb, rd] = -(np.asarray(force_bar, dtype=float) * self.force_multiplier)[:, None] * inertia[rb, rd][None]
        y_bar, g = self._force.engine.rhs_vjp(y, 0.0, lam, which=("k_bond",))
        fields_bar = np.zeros_like(solution_data.fields)
        fields_bar[:, 0] = y_bar[:, 0]
        return fields_bar, g["k_bond"].sum((0, 1))


@dataclass
class HingeForward(_HingeBase):
    """``problems/hinge_characterization.py:ForwardProblem`` (fields with the same names): static tests on rotating-square samples."""
    n1_cells: int
    n2_cells: int
    spacing: Any
    bond_length: Any
    initial_angle: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    loading_type: str
    amplitude: Any
    loading_rate: Any
    n_timepoints: int
    linearized_strains: bool = False
    force_multiplier: float = 1.
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    atol: float = 1e-8
    rtol: float = 1e-8
    steps_per_interval: Optional[int] = None
    device: int = 0
    name: str = "hinge_characterization"
    _lib: InitVar[Any] = None      # test infrastructure only

    def __post_init__(self, _lib=None):
        self._lib, self.is_setup = _lib, False

    def _make_geometry(self):
        return RotatedSquareGeometry(self.n1_cells, self.n2_cells, self.spacing, self.bond_length), (self.initial_angle,)

    def to_dict(self):
        import dataclasses
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

    @classmethod
    def from_dict(cls, dict_in, _lib=None):
        return cls(**dict_in, _lib=_lib)


@dataclass
class HingeQuadsForward(_HingeBase):
    """``problems/hinge_characterization.py:ForwardProblemQuads``: validation of the hinge model on a random quad sample."""
    n1_blocks: int
    n2_blocks: int
    spacing: Any
    bond_length: Any
    horizontal_shifts: Any
    vertical_shifts: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    loading_type: str
    amplitude: Any
    loading_rate: Any
    n_timepoints: int
    linearized_strains: bool = False
    force_multiplier: float = 1.
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    atol: float = 1e-8
    rtol: float = 1e-8
    steps_per_interval: Optional[int] = None
    device: int = 0
    name: str = "hinge_characterization"
    _lib: InitVar[Any] = None

    def __post_init__(self, _lib=None):
        self._lib, self.is_setup = _lib, False

    def _make_geometry(self):
        return QuadGeometry(self.n1_blocks, self.n2_blocks, self.spacing, self.bond_length), (self.horizontal_shifts, self.vertical_shifts)

    to_dict = HingeForward.to_dict
    from_dict = classmethod(HingeForward.from_dict.__func__)


def naive_GD(value_and_grad, initial_guess, step_size, n_iterations, lower_bound=None, upper_bound=None, verbose=False):
    """Gradient descent with a fixed step and bounds (hinge_characterization.py:563-585); ``value_and_grad(design) -> (value, grad)``
    stands where the reference jit-compiles ``value_and_grad(objective_fn)``."""
    obj_values, design_values = [], [tuple(initial_guess)]
    for i in range(n_iterations):
        value, update = value_and_grad(design_values[i])
        obj_values.append(value)
        design_values.append(tuple(float(enforce_bounds(x - step_size * dx, lower_bound, upper_bound)) for x, dx in zip(design_values[i], update)))
        if verbose:
            print(f"Objective = {value:.9f}")
    return obj_values, design_values


@dataclass
class HingeResponseError:
    """``problems/hinge_characterization.py:OptimizationProblem``: the stiffnesses that make the simulated force-displacement curves of
    several tests (one forward problem per loading type) match the measured ones.  ``target_responses[loading_type]`` is
    ``[displacement_history, force_history, force_std]``."""
    forward_problems: List[Any]
    target_responses: Dict[str, Any]
    fitted_responses: Optional[Dict[str, Any]] = None
    objective_values: Optional[list] = None
    design_values: Optional[list] = None
    name: str = "hinge_characterization"

    def __post_init__(self):
        self.objective_values = [] if self.objective_values is None else self.objective_values
        self.design_values = [] if self.design_values is None else self.design_values
        self.is_setup = False

    def setup_objective(self):
        for p in self.forward_problems:
            if not p.is_setup:
                p.setup()
        n = self.forward_problems[0].n_timepoints          # targets and simulations sampled alike (:631-635): the ramp is linear
        self.target_forces = np.array([resample(u, f, n) for u, f, _ in self.target_responses.values()])
        self.is_setup = True

    def compute_fitted_responses(self, k_values):
        for p in self.forward_problems:
            if not p.is_setup:
                p.setup()
        return {p.loading_type: p.force_displacement(*p.solve(k_values)) for p in self.forward_problems}

    def objective_fn(self, k_values):
        """mean((reaction forces - target forces)^2) over tests and output times (:637-646)."""
        if not self.is_setup:
            self.setup_objective()
        forces = np.array([f for _, f in self.compute_fitted_responses(k_values).values()])
        return float(np.mean((forces - self.target_forces) ** 2))

    def value_and_grad(self, k_values):
        if not self.is_setup:
            self.setup_objective()
        total = self.target_forces.size
        value, grad = 0.0, np.zeros(3)
        for p, target in zip(self.forward_problems, self.target_forces):
            sol, cp = p.solve(k_values, keep_trajectory=True)
            _, forces = p.force_displacement(sol, cp)
            value += float(((forces - target) ** 2).sum()) / total
            fields_bar, explicit = p.force_vjp(sol, cp, 2.0 * (forces - target) / total)
            implicit = p.solve_dynamics.vjp_raw(fields_bar, which=("k_bond",))["k_bond"]
            grad += explicit + np.asarray(implicit, dtype=float)[0].sum(0)
        return value, tuple(grad)

    def run_optimization_GD(self, initial_guess, n_iterations, step_size, lower_bound=None, upper_bound=None, verbose=False):
        self.objective_values, self.design_values = naive_GD(self.value_and_grad, initial_guess, step_size, n_iterations, lower_bound,
                                                             upper_bound, verbose)
        self.fitted_responses = self.compute_fitted_responses(self.design_values[-1])

    def run_optimization_nlopt(self, initial_guess, n_iterations, max_time=None, lower_bound=None, upper_bound=None, verbose=False):
        """:668-721 with the method of moving asymptotes of ``difflexmm_amd.optimize`` where the reference calls NLopt's LD_MMA
        (minimisation, bounds, at most ``n_iterations`` evaluations)."""
        import time
        from .optimize import mma_minimize
        t0 = time.perf_counter()

        class _TimeUp(Exception):
            pass

        def fun(x):
            if max_time is not None and self.objective_values and time.perf_counter() - t0 > max_time:
                raise _TimeUp
            v, g = self.value_and_grad(tuple(x))
            self.objective_values.append(v)
            self.design_values.append(tuple(float(a) for a in x))
            if verbose:
                print(f"Iteration: {len(self.objective_values)}\nObjective = {v}")
            return v, np.asarray(g, dtype=float)
        try:
            mma_minimize(fun, np.asarray(initial_guess, dtype=float), lower=lower_bound, upper=upper_bound, maxeval=n_iterations)
        except _TimeUp:
            pass
        self.fitted_responses = self.compute_fitted_responses(self.design_values[-1])

    def to_dict(self):
        return dict(forward_problems=[p.to_dict() for p in self.forward_problems], target_responses=self.target_responses,
                    fitted_responses=self.fitted_responses, objective_values=list(self.objective_values),
                    design_values=list(self.design_values), name=self.name)

    @staticmethod
    def from_dict(dict_in, _lib=None):
        d = dict(dict_in)
        d["forward_problems"] = [(HingeQuadsForward if "n1_blocks" in p else HingeForward).from_dict(p, _lib=_lib) for p in d["forward_problems"]]
        return HingeResponseError(**d)
