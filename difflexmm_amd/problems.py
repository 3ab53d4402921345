"""Boundary-condition patterns, target-block lists and objectives of the reference's focusing problems
(the callers of the hot path), restated in NumPy:

* quads:  ``problems/quads_focusing.py:104-209`` (driven edge blocks + clamped corners), ``:447-451`` (target)
* kagome: ``problems/kagome_focusing.py:96-172``, ``:403-407``
* objective ``target_kinetic_energy``: ``problems/quads_focusing.py:453-467`` with ``energy.py:494-499``
"""
from dataclasses import InitVar, dataclass
from typing import Any, Optional, Tuple

import os

import numpy as np

from . import energy as E
from . import loading as L
from .dynamics import setup_dynamic_solver
from .geometry import KagomeGeometry, QuadGeometry, compute_inertia
from .utils import (ContactParams, ControlParams, GeometricalParams, LigamentParams, MechanicalParams, SolutionData)


_GEOMETRY_CACHE = {}   # (lattice signature, ids of the design arrays) -> (design arrays, block_centroids, centroid_node_vectors)


def _signature(geometry):
    sig = getattr(geometry, "_signature", None)
    if sig is None:
        # every public attribute that defines the lattice: scalars by value, arrays (kagome direct_basis) by content
        sig = geometry._signature = (type(geometry).__name__,) + tuple(
            sorted((k, float(v) if isinstance(v, (int, float)) else (np.asarray(v).shape, np.asarray(v, dtype=float).tobytes()))
                   for k, v in vars(geometry).items()
                   if isinstance(v, (int, float, np.ndarray, list, tuple)) and not k.startswith("_")))
    return sig


def _remember_geometry(geometry, design, centroids, cnv):
    centroids.flags.writeable = False
    cnv.flags.writeable = False
    if len(_GEOMETRY_CACHE) > 1024:
        _GEOMETRY_CACHE.clear()
    _GEOMETRY_CACHE[(_signature(geometry),) + tuple(id(a) for a in design)] = (tuple(design), centroids, cnv)   # (holds the design arrays: ids stay unique)


def geometry_from_design_cached(geometry, design):
    """``geometry.geometry_from_design(*design)`` remembered per design OBJECT: the forward problems of a multi-input
    objective (one solver per input) all ask for the geometry of the same design tuple in the same round."""
    key = (_signature(geometry),) + tuple(id(a) for a in design)
    hit = _GEOMETRY_CACHE.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[0], design)):
        return hit[1], hit[2]
    centroids, cnv = geometry.geometry_from_design(*design)
    _remember_geometry(geometry, design, centroids, cnv)
    return centroids, cnv


def _native_design_map(fw):
    """(library, NativeDesignMap) of a forward problem whose lattice map runs in native code (include/dfx.h: dfx_design_forward / _vjp), or None:
    a scalar density, a lattice class with ``design_shapes`` / ``reference_node_vectors``, a library that exports the two entry points."""
    hit = getattr(fw, "_native_map", None)
    if hit is not None:
        return hit or None
    fw._native_map = False
    try:
        if np.ndim(fw.density) != 0 or not hasattr(fw.geometry, "design_shapes") or os.environ.get("DFX_NATIVE_DESIGN", "1") == "0":
            return None
        from ._binding import load_library
        from .geometry import NativeDesignMap
        lib = fw._lib if getattr(fw, "_lib", None) is not None else load_library()
        if not hasattr(lib, "dfx_design_forward"):
            return None
        spec_contact = getattr(fw.solve_dynamics.spec, "contact", 0)
        fw._native_map = (lib, NativeDesignMap(fw.geometry, fw.solve_dynamics.bonds), spec_contact)
    except Exception:       # noqa: BLE001 -- the NumPy maps are always there
        return None
    return fw._native_map


def prefetch_designs(fw, designs):
    """The geometry of every design of a batch in ONE native pass (lattice map + polygon pass + inertia + undeformed void angles, fused:
    csrc/dfx_design.h), left where ``control_params`` / ``DynamicSolver._flatten`` look for it -- the geometry cache above and the solver's
    cache of what it derives from a design.  No-op when the native map does not apply or everything is cached already."""
    nm = _native_design_map(fw)
    if nm is None:
        return
    lib, ndm, contact = nm
    sig = _signature(fw.geometry)
    todo = []
    for d in designs:
        hit = _GEOMETRY_CACHE.get((sig,) + tuple(id(a) for a in d))
        if not (hit is not None and all(a is b for a, b in zip(hit[0], d))):
            todo.append(d)
    if not todo:
        return
    from . import _binding as _b
    from .dynamics import remember_flat
    cen, cnv, inertia, va = ndm.forward(lib, todo, float(fw.density), void_angles=contact == _b.CONTACT_ANGLE)
    for i, d in enumerate(todo):
        ci, vi, ii, ai = cen[i], cnv[i], inertia[i], (None if va is None else va[i])
        ii.flags.writeable = False
        if ai is not None:
            ai.flags.writeable = False
        _remember_geometry(fw.geometry, d, ci, vi)
        remember_flat(vi, fw.solve_dynamics.bonds, fw.density, ii, ai)


def _tile3(blocks):
    blocks = np.asarray(blocks, dtype=np.int64)
    n = len(blocks)
    return np.stack([np.tile(blocks, 3), np.repeat(np.arange(3), n)], 1)


def quads_focusing_constraints(geometry: QuadGeometry, n_excited_blocks: int, loaded_side: str = "left",
                               input_shift: int = 0, n_blocks_clamped_corner: int = 2):
    """(constrained_block_DOF_pairs, loading_vector, driven_blocks, clamped_blocks) exactly as
    problems/quads_focusing.py:104-209 builds them (same ordering of the pairs)."""
    n1, n2, nb = geometry.n1_blocks, geometry.n2_blocks, geometry.n_blocks
    ne, sh, nc = n_excited_blocks, input_shift, n_blocks_clamped_corner
    if loaded_side == "left":
        blocks = np.arange((n2 - ne) // 2 + sh, (n2 + ne) // 2 + sh) * n1
        dofs = [0] * ne + [1] * ne + [2] * ne
    elif loaded_side == "right":
        blocks = np.arange((n2 - ne) // 2 + sh, (n2 + ne) // 2 + sh) * n1 + (n1 - 1)
        dofs = [0] * ne + [1] * ne + [2] * ne
    elif loaded_side == "bottom":
        blocks = np.arange((n1 - ne) // 2 + sh, (n1 + ne) // 2 + sh)
        dofs = [1] * ne + [0] * ne + [2] * ne
    elif loaded_side == "top":
        blocks = np.arange((n1 - ne) // 2 + sh, (n1 + ne) // 2 + sh) + n1 * (n2 - 1)
        dofs = [1] * ne + [0] * ne + [2] * ne
    else:
        raise ValueError(f"Unknown loaded_side: {loaded_side}. Should be either 'left', 'right', 'bottom' or 'top'.")
    driven = np.stack([np.tile(blocks, 3), np.array(dofs)], 1)
    bl = np.concatenate([np.arange(0, nc), [i * n1 for i in range(1, nc)]])
    br = np.concatenate([np.arange(n1 - nc, n1), [(i + 1) * n1 - 1 for i in range(1, nc)]])
    tr = np.concatenate([np.arange(nb - nc, nb), [nb - i * n1 - 1 for i in range(1, nc)]])
    tl = np.concatenate([np.arange(nb - n1, nb - n1 + nc), [nb - n1 - i * n1 for i in range(1, nc)]])
    clamped = [_tile3(c) for c in (bl, br, tr, tl)]
    pairs = np.concatenate([driven] + clamped).astype(np.int64)
    vec = np.zeros(len(pairs))
    vec[:ne] = 1.0
    return pairs, vec, np.unique(driven[:, 0]), np.unique(np.concatenate(clamped)[:, 0])


def quads_target_blocks(geometry: QuadGeometry, target_size: Tuple[int, int], target_shift: Tuple[int, int]):
    """problems/quads_focusing.py:447-451."""
    n1, n2 = geometry.n1_blocks, geometry.n2_blocks
    return np.array([j * n1 + i
                     for i in range((n1 - target_size[0]) // 2 + target_shift[0], (n1 + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2 - target_size[1]) // 2 + target_shift[1], (n2 + target_size[1]) // 2 + target_shift[1])],
                    dtype=np.int32)


def kagome_focusing_constraints(geometry: KagomeGeometry, n_excited_blocks: int, n_blocks_clamped_corner: int = 2):
    """problems/kagome_focusing.py:96-160 (left-loaded only, as in the reference)."""
    n1, n2, ncell = geometry.n1_cells, geometry.n2_cells, geometry.n_cells
    ne, nc = n_excited_blocks, n_blocks_clamped_corner
    blocks = np.arange(2 * n1 * ((n2 - ne) // 2), 2 * n1 * ((n2 + ne) // 2), 2 * n1)
    driven = np.stack([np.tile(blocks, 3), np.array([0] * ne + [1] * ne + [2] * ne)], 1)
    bl = np.concatenate([np.arange(0, nc), [i * n1 for i in range(1, nc)]]) * 2
    br = np.concatenate([np.arange(n1 - nc, n1) * 2, [(i + 1) * 2 * n1 - 1 for i in range(0, nc)]])
    tr = np.concatenate([np.arange(ncell - nc, ncell), [ncell - i * n1 - 1 for i in range(1, nc)]]) * 2 + 1
    tl = np.concatenate([np.arange(ncell - n1, ncell - n1 + nc) * 2 + 1,
                         np.array([ncell - n1 - i * n1 for i in range(0, nc)]) * 2])
    clamped = [_tile3(c) for c in (bl, br, tr, tl)]
    pairs = np.concatenate([driven] + clamped).astype(np.int64)
    vec = np.zeros(len(pairs))
    vec[:ne] = 1.0
    return pairs, vec, np.unique(driven[:, 0]), np.unique(np.concatenate(clamped)[:, 0])


def kagome_target_blocks(geometry: KagomeGeometry, target_size, target_shift):
    """problems/kagome_focusing.py:403-407."""
    n1, n2 = geometry.n1_cells, geometry.n2_cells
    return np.array([(2 * (j * n1 + i), 2 * (j * n1 + i) + 1)
                     for i in range((n1 - target_size[0]) // 2 + target_shift[0], (n1 + target_size[0]) // 2 + target_shift[0])
                     for j in range((n2 - target_size[1]) // 2 + target_shift[1], (n2 + target_size[1]) // 2 + target_shift[1])],
                    dtype=np.int32).flatten()


def _make_drive(self, vec, excited_blocks_fn):
    """The synthetic pulse (harmonic signal for the spin problem) on the driven DOFs, or -- ``setup(excited_blocks_fn=...)`` of the
    reference's forward problems, e.g. problems/quads_focusing.py:213-222: "user-defined loading, can be used to apply the experimental
    loading" -- a recorded signal.  The reference takes any callable of t; on the device it is a ``loading.Table`` (piecewise linear,
    end values held: how the notebooks build theirs with ``jnp.interp``)."""
    self._recorded = excited_blocks_fn is not None
    if excited_blocks_fn is None:
        return self._drive_cls(vec)
    if not isinstance(excited_blocks_fn, L.Table):
        raise TypeError("excited_blocks_fn must be a difflexmm_amd.loading.Table (a recorded signal the device can interpolate)")
    return L.Table(excited_blocks_fn.times, excited_blocks_fn.values, vector=vec)


def _drive_params(self):
    if getattr(self, "_recorded", False):
        return dict(amplitude=1.0, delay=0.0)
    return dict(amplitude=self.signed_amplitude, loading_rate=self.loading_rate, input_delay=self.input_delay)


@dataclass
class QuadsFocusingForward:
    """NumPy counterpart of ``problems/quads_focusing.py:ForwardProblem`` (fields with the same names)."""
    n1_blocks: int
    n2_blocks: int
    spacing: Any
    bond_length: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    amplitude: Any
    loading_rate: Any
    input_delay: Any
    n_excited_blocks: int
    loaded_side: str
    input_shift: int
    simulation_time: Any
    n_timepoints: int
    linearized_strains: bool = False
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    n_blocks_clamped_corner: int = 2
    atol: float = 1e-8              # solver tolerances (problems/quads_focusing.py:73-74): the adaptive controller's, and those of
    rtol: float = 1e-8              # the grid it freezes for gradients, when steps_per_interval is None
    steps_per_interval: Optional[int] = None
    grid_refine: int = 1            # gradients on the frozen adaptive grid: every accepted step split into this many (DynamicSolver.adaptive_grid)
    integrator: str = "dopri5"
    batch: int = 1
    device: int = 0
    streams: int = 0          # member groups on their own HIP streams (0: the engine chooses); multi-input objectives set 1
    name: str = "quads_focusing"
    _lib: InitVar[Any] = None      # test infrastructure only (the CPU port of the oracle): not a field, never serialised

    def __post_init__(self, _lib=None):
        self._lib = _lib
    _drive_cls = L.Pulse          # problems/quads_focusing.py:211-222

    def _make_geometry(self):
        return QuadGeometry(self.n1_blocks, self.n2_blocks, self.spacing, self.bond_length)

    def setup(self, excited_blocks_fn=None):
        """problems/quads_focusing.py:82-317 (and, through ``_make_geometry``, problems/reference_design.py): geometry, boundary
        conditions, energy, solver.  ``excited_blocks_fn``: a recorded input signal (``loading.Table``) instead of the synthetic pulse."""
        g = self.geometry = self._make_geometry()
        self.bond_connectivity = g.bond_connectivity()
        self.reference_bond_vectors = g.reference_bond_vectors()
        pairs, vec, self.driven_blocks_ids, self.clamped_blocks_ids = quads_focusing_constraints(
            g, self.n_excited_blocks, self.loaded_side, self.input_shift, self.n_blocks_clamped_corner)
        self.constrained_block_DOF_pairs = pairs
        self.moving_blocks_ids = np.setdiff1d(np.arange(g.n_blocks), self.clamped_blocks_ids)
        strain = E.build_strain_energy(self.bond_connectivity,
                                       E.ligament_energy_linearized if self.linearized_strains else E.ligament_energy)
        energy = E.combine_block_energies(strain, E.build_contact_energy(self.bond_connectivity)) if self.use_contact else strain
        self.solve_dynamics = setup_dynamic_solver(
            g, energy, constrained_block_DOF_pairs=pairs, constrained_DOFs_fn=_make_drive(self, vec, excited_blocks_fn),
            damped_blocks=np.arange(g.n_blocks), rtol=self.rtol, atol=self.atol, integrator=self.integrator,
            steps_per_interval=self.steps_per_interval, batch=self.batch, device=self.device, streams=self.streams,
            grid_refine=getattr(self, "grid_refine", 1), _lib=self._lib)
        self.timepoints = np.linspace(0, self.simulation_time, self.n_timepoints)
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.signed_amplitude = self.amplitude if self.loaded_side in ("left", "bottom") else -self.amplitude
        self.is_setup = True

    def control_params(self, design):
        prefetch_designs(self, [design])           # (native lattice map when it applies: geometry, inertia, void angles in one pass)
        centroids, cnv = geometry_from_design_cached(self.geometry, design)
        return ControlParams(
            geometrical_params=GeometricalParams(block_centroids=centroids, centroid_node_vectors=cnv),
            mechanical_params=MechanicalParams(
                bond_params=LigamentParams(self.k_stretch, self.k_shear, self.k_rot, self.reference_bond_vectors),
                density=self.density, damping=self.damping,
                contact_params=ContactParams(min_angle=self.min_angle, cutoff_angle=self.cutoff_angle, k_contact=self.k_contact)),
            constraint_params=_drive_params(self))

    def solve(self, design, keep_trajectory=False, want_fields=True):
        """design = (horizontal_shifts, vertical_shifts), or a list of ``batch`` such tuples.
        want_fields=False leaves the histories on the device (for objectives the engine evaluates there) and returns None."""
        many = isinstance(design, list)
        if many:
            prefetch_designs(self, design)         # all designs of the batch in one native pass
        cps = [self.control_params(d) for d in design] if many else self.control_params(design)
        fields = self.solve_dynamics(self.state0, self.timepoints, cps, keep_trajectory=keep_trajectory, want_fields=want_fields)
        self._last_design = design
        if fields is None:
            return None
        if many:
            sols = [SolutionData(cp.geometrical_params.block_centroids, cp.geometrical_params.centroid_node_vectors,
                                 self.bond_connectivity, self.timepoints, f) for cp, f in zip(cps, fields)]
        else:
            sols = [SolutionData(cps.geometrical_params.block_centroids, cps.geometrical_params.centroid_node_vectors,
                                 self.bond_connectivity, self.timepoints, fields)]
        # remembered so that compute_response_data can reduce THIS solve's histories on the device
        self._last_solutions, self._last_solve_id, self.solution_data = sols, self.solve_dynamics.solve_count, sols[0]
        return sols if many else sols[0]


@dataclass
class RotatedSquaresForward(QuadsFocusingForward):
    """``problems/reference_design.py:ForwardProblem``: the regular rotated-squares domain (``RotatedSquareGeometry`` of
    n1_blocks/2 x n2_blocks/2 cells at ``initial_angle``) with the boundary conditions and the pulse of the focusing problems -- the
    design the optimised lattices are compared with.  ``solve()`` takes no design; ``setup(excited_blocks_fn=...)`` replaces the
    synthetic pulse by a recorded signal (a ``loading.Table``, the reference's ``excited_blocks_fn(t)``, reference_design.py:206-216)."""
    initial_angle: Any = None
    name: str = "rotated_squares"

    def _make_geometry(self):
        from .geometry import RotatedSquareGeometry
        if self.initial_angle is None:
            raise ValueError("RotatedSquaresForward needs initial_angle")
        return RotatedSquareGeometry(self.n1_blocks // 2, self.n2_blocks // 2, self.spacing, self.bond_length)

    def control_params(self, design=None):
        g = self.geometry
        return ControlParams(
            geometrical_params=GeometricalParams(block_centroids=g.block_centroids(self.initial_angle),
                                                 centroid_node_vectors=g.centroid_node_vectors(self.initial_angle)),
            mechanical_params=MechanicalParams(
                bond_params=LigamentParams(self.k_stretch, self.k_shear, self.k_rot, self.reference_bond_vectors),
                density=self.density, damping=self.damping,
                contact_params=ContactParams(min_angle=self.min_angle, cutoff_angle=self.cutoff_angle, k_contact=self.k_contact)),
            constraint_params=_drive_params(self))

    def solve(self, design=None, keep_trajectory=False, want_fields=True):
        cp = self.control_params()
        fields = self.solve_dynamics(self.state0, self.timepoints, cp, keep_trajectory=keep_trajectory, want_fields=want_fields)
        if fields is None:
            return None
        sol = SolutionData(cp.geometrical_params.block_centroids, cp.geometrical_params.centroid_node_vectors, self.bond_connectivity,
                           self.timepoints, fields)
        self._last_solutions, self._last_solve_id, self.solution_data = [sol], self.solve_dynamics.solve_count, sol
        return sol


def _compute_response_data(self, solution_data=None):
    """Strain-energy and kinetic-energy histories of a solution (problems/quads_focusing.py:319-372): the fields of
    SolutionData plus strain_energy_{stretch,shear,bending} (T, n_bonds) and kinetic_energy (T, n_blocks).
    For the solution(s) of the LAST solve -- ``solution_data`` omitted, as in the reference, or one of the objects that solve
    returned -- the histories are reduced on the device from the resident fields (``dfx_response_data``); any other
    SolutionData goes through the NumPy formulas on the host."""
    last = getattr(self, "_last_solutions", None)
    if solution_data is None:
        if not last:
            raise ValueError("No solution data available!")
        solution_data = last[0]
    if type(solution_data) is not SolutionData:
        raise ValueError("Solution data is not of type SolutionData!")
    out = solution_data._asdict()
    member = next((i for i, s_ in enumerate(last or []) if s_ is solution_data), None)
    if member is not None and getattr(self, "_last_solve_id", None) == self.solve_dynamics.solve_count:
        cache = getattr(self, "_response_cache", None)
        if cache is None or cache[0] is not last:
            cache = self._response_cache = (last, self.solve_dynamics.engine.response_data())
        for k, v in cache[1].items():
            out[k] = v[member]
        return out
    axial, shear, bending = E.compute_ligament_strains_history(solution_data.fields[:, 0], solution_data.centroid_node_vectors,
                                                               solution_data.bond_connectivity, self.reference_bond_vectors)
    out["strain_energy_stretch"] = 0.5 * self.k_stretch * (axial * self.bond_length) ** 2
    out["strain_energy_shear"] = 0.5 * self.k_shear * (shear * self.bond_length) ** 2
    out["strain_energy_bending"] = 0.5 * self.k_rot * bending ** 2
    inertia = compute_inertia(solution_data.centroid_node_vectors, self.density)
    out["kinetic_energy"] = np.sum(0.5 * solution_data.fields[:, 1] ** 2 * inertia, axis=-1)
    return out


def _forward_to_dict(self):
    """problems/quads_focusing.py:398-407: the dataclass fields (construction arguments) plus ``solution_data`` with the
    SolutionData namedtuple(s) turned into plain dicts, ready for ``utils.save_data``.  Runtime handles (engine, library)
    are not part of it."""
    import dataclasses
    out = {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if not f.name.startswith("_")}
    sol = getattr(self, "solution_data", None)
    many = getattr(self, "_last_solutions", None)
    if many is not None and len(many) > 1 and sol is many[0]:
        out["solution_data"] = [s_._asdict() for s_ in many]
    else:
        out["solution_data"] = sol._asdict() if isinstance(sol, SolutionData) else None
    return out


def _forward_from_dict(cls, dict_in, _lib=None):
    """problems/quads_focusing.py:384-396: rebuild the problem (not set up yet) and turn stored solutions back into
    SolutionData."""
    d = dict(dict_in)
    sol = d.pop("solution_data", None)
    fw = cls(**d, _lib=_lib)
    if isinstance(sol, dict):
        fw.solution_data = SolutionData(**sol)
    elif isinstance(sol, list):
        fw._stored_solutions = [SolutionData(**s_) for s_ in sol]
        fw.solution_data = fw._stored_solutions[0]
    else:
        fw.solution_data = None
    fw.is_setup = False
    return fw


def _forward_to_data(self):
    """problems/quads_focusing.py:378-379 (``to_data``): a copy without the solver -- what can be pickled."""
    return type(self).from_dict(self.to_dict(), _lib=getattr(self, "_lib", None))


def _forward_from_data(cls, problem_data, _lib=None):
    """problems/quads_focusing.py:372-376 (``from_data``): from a stored problem object (or its dict), not set up."""
    return cls.from_dict(problem_data if isinstance(problem_data, dict) else problem_data.to_dict(), _lib=_lib)


QuadsFocusingForward.compute_response_data = _compute_response_data
QuadsFocusingForward.to_dict = _forward_to_dict
QuadsFocusingForward.from_dict = classmethod(_forward_from_dict)
QuadsFocusingForward.to_data = _forward_to_data
QuadsFocusingForward.from_data = classmethod(_forward_from_data)


@dataclass
class QuadsSpinForward(QuadsFocusingForward):
    """``problems/quads_spin.py:ForwardProblem``: the focusing problem's lattice and boundary conditions driven by the harmonic
    signal of ``quads_spin.py:210-222`` (the raised cosine switched on at ``t > input_delay`` and never off)."""
    name: str = "quads_spin"
    _drive_cls = L.Harmonic


@dataclass
class KagomeFocusingForward:
    """NumPy counterpart of ``problems/kagome_focusing.py:ForwardProblem`` (fields with the same names; left-loaded)."""
    n1_cells: int
    n2_cells: int
    cell_size: Any
    bond_length: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    amplitude: Any
    loading_rate: Any
    input_delay: Any
    n_excited_blocks: int
    simulation_time: Any
    n_timepoints: int
    loaded_side: str = "left"
    cell_angle: Any = np.pi / 3
    linearized_strains: bool = False
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    n_blocks_clamped_corner: int = 2
    atol: float = 1e-8              # solver tolerances (problems/quads_focusing.py:73-74): the adaptive controller's, and those of
    rtol: float = 1e-8              # the grid it freezes for gradients, when steps_per_interval is None
    steps_per_interval: Optional[int] = None
    integrator: str = "dopri5"
    batch: int = 1
    device: int = 0
    streams: int = 0
    name: str = "kagome_focusing"
    _lib: InitVar[Any] = None      # test infrastructure only (the CPU port of the oracle): not a field, never serialised

    def __post_init__(self, _lib=None):
        self._lib = _lib

    _drive_cls = L.Pulse

    def setup(self, excited_blocks_fn=None):
        if self.loaded_side != "left":
            raise ValueError(f"Unknown loaded_side: {self.loaded_side}. Only 'left' is implemented.")   # kagome_focusing.py:107-109
        basis = self.cell_size * np.array([[1.0, 0.0], [np.cos(self.cell_angle), np.sin(self.cell_angle)]])
        g = self.geometry = KagomeGeometry(self.n1_cells, self.n2_cells, basis, self.bond_length)
        self.bond_connectivity = g.bond_connectivity()
        self.reference_bond_vectors = g.reference_bond_vectors()
        pairs, vec, self.driven_blocks_ids, self.clamped_blocks_ids = kagome_focusing_constraints(
            g, self.n_excited_blocks, self.n_blocks_clamped_corner)
        self.constrained_block_DOF_pairs = pairs
        self.moving_blocks_ids = np.setdiff1d(np.arange(g.n_blocks), self.clamped_blocks_ids)
        strain = E.build_strain_energy(self.bond_connectivity,
                                       E.ligament_energy_linearized if self.linearized_strains else E.ligament_energy)
        energy = E.combine_block_energies(strain, E.build_contact_energy(self.bond_connectivity)) if self.use_contact else strain
        self.solve_dynamics = setup_dynamic_solver(
            g, energy, constrained_block_DOF_pairs=pairs, constrained_DOFs_fn=_make_drive(self, vec, excited_blocks_fn),
            damped_blocks=np.arange(g.n_blocks), rtol=self.rtol, atol=self.atol, integrator=self.integrator,
            steps_per_interval=self.steps_per_interval, batch=self.batch, device=self.device, streams=self.streams,
            grid_refine=getattr(self, "grid_refine", 1), _lib=self._lib)
        self.timepoints = np.linspace(0, self.simulation_time, self.n_timepoints)
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.signed_amplitude = self.amplitude
        self.is_setup = True

    control_params = QuadsFocusingForward.control_params
    solve = QuadsFocusingForward.solve
    compute_response_data = _compute_response_data
    to_dict = _forward_to_dict
    from_dict = classmethod(_forward_from_dict)
    to_data = _forward_to_data
    from_data = classmethod(_forward_from_data)


def static_tuning_constraints(geometry: QuadGeometry, n_excited_blocks: int, input_shift: int = 0):
    """``problems/quads_kinetic_energy_static_tuning.py:124-170``: driven blocks centred on the left edge (x driven, y / theta
    held), bottom and top rows clamped in (y, x, theta) order; the dynamic loading vector selects the driven x DOFs, the static
    one the y DOFs of the bottom (+1/2) and top (-1/2) rows.  Returns (pairs, dynamic vector, static vector, driven ids,
    clamped ids)."""
    n1, n2, nb, ne = geometry.n1_blocks, geometry.n2_blocks, geometry.n_blocks, n_excited_blocks
    driven = np.stack([np.tile(np.arange((n2 - ne) // 2 + input_shift, (n2 + ne) // 2 + input_shift) * n1, 3),
                       np.array([0] * ne + [1] * ne + [2] * ne)], 1)
    order = np.array([1] * n1 + [0] * n1 + [2] * n1)
    bottom = np.stack([np.concatenate([np.arange(0, n1)] * 3), order], 1)
    top = np.stack([np.concatenate([np.arange(nb - n1, nb)] * 3), order], 1)
    pairs = np.concatenate([driven, bottom, top]).astype(np.int64)
    dyn = np.zeros(len(pairs))
    dyn[:ne] = 1
    sta = np.zeros(len(pairs))
    sta[3 * ne:3 * ne + n1] = 0.5
    sta[3 * ne + 3 * n1:3 * ne + 4 * n1] = -0.5
    clamped = np.unique(np.concatenate([bottom, top])[:, 0])
    return pairs, dyn, sta, np.unique(driven[:, 0]), clamped


@dataclass
class ForwardInput:
    """``quads_kinetic_energy_static_tuning.py:ForwardInput``: one entry per forward problem (the rows the reference maps over
    devices with ``pmap``) in each of the four loading tuples."""
    horizontal_shifts: Any
    vertical_shifts: Any
    amplitude: Tuple[Any, ...]
    loading_rate: Tuple[Any, ...]
    compressive_strain: Tuple[Any, ...]
    compressive_strain_rate: Tuple[Any, ...]

    def rows(self):
        return np.array([self.amplitude, self.loading_rate, self.compressive_strain, self.compressive_strain_rate], dtype=float).T


@dataclass
class QuadsStaticTuningForward:
    """NumPy counterpart of ``problems/quads_kinetic_energy_static_tuning.py:ForwardProblem`` (fields with the same names):
    bottom and top edges clamped and slowly compressed (``compressive_strain`` at ``compressive_strain_rate``), one pulse on the
    left edge once the compression has ended.

    The reference maps the forward inputs over devices (``pmap``, ``:473-478``).  Here the rows are ensemble members of ONE engine
    call: a member's time grid starts with its own static phase ``[0, strain / strain_rate + input_delay]``, so rows with different
    strains or rates have different grids -- ``dfx_forward_grid_members`` integrates every member on its own times with shared step
    counts (``static_steps`` RK steps in the static interval, ``steps_per_interval`` between dynamic outputs).  With both None the
    adaptive controller chooses the grid (and freezes it for gradients) per call: then only rows with equal output times share one,
    the others run one after the other (``ensemble`` shards rows over ranks)."""
    n1_blocks: int
    n2_blocks: int
    spacing: Any
    bond_length: Any
    k_stretch: Any
    k_shear: Any
    k_rot: Any
    density: Any
    damping: Any
    n_excited_blocks: int
    input_shift: int
    simulation_time_dynamic: Any
    n_timepoints: int
    linearized_strains: bool = False
    use_contact: bool = True
    k_contact: Any = 1.
    min_angle: Any = 0. * np.pi / 180
    cutoff_angle: Any = 5. * np.pi / 180
    name: str = "quads_kinetic_energy_static_tuning"
    atol: float = 1e-8
    rtol: float = 1e-8
    steps_per_interval: Optional[int] = None
    static_steps: Optional[int] = None
    integrator: str = "dopri5"
    device: int = 0
    streams: int = 0
    _lib: InitVar[Any] = None      # test infrastructure only (the CPU port of the oracle): not a field, never serialised

    def __post_init__(self, _lib=None):
        self._lib = _lib

    def setup(self):
        g = self.geometry = QuadGeometry(self.n1_blocks, self.n2_blocks, self.spacing, self.bond_length)
        self.bond_connectivity = g.bond_connectivity()
        self.reference_bond_vectors = g.reference_bond_vectors()
        pairs, dyn, sta, self.driven_blocks_ids, self.clamped_blocks_ids = static_tuning_constraints(
            g, self.n_excited_blocks, self.input_shift)
        self.constrained_block_DOF_pairs = pairs
        self.constrained_DOFs_loading_vector_dynamic, self.constrained_DOFs_loading_vector_static = dyn, sta
        self.moving_blocks_ids = np.setdiff1d(np.arange(g.n_blocks), self.clamped_blocks_ids)
        strain = E.build_strain_energy(self.bond_connectivity,
                                       E.ligament_energy_linearized if self.linearized_strains else E.ligament_energy)
        self._energy = E.combine_block_energies(strain, E.build_contact_energy(self.bond_connectivity)) if self.use_contact else strain
        self._drive = L.static_tuning_drive(sta, dyn, (g.n2_blocks - 1) * self.spacing)       # :176-186
        self._solvers = {}
        self.state0 = np.zeros((2, g.n_blocks, 3))
        self.solve_dynamics = self.solver(1)
        self.is_setup = True

    def solver(self, batch):
        """The solver that integrates ``batch`` rows side by side (built on first use)."""
        sd = self._solvers.get(batch)
        if sd is None:
            sd = self._solvers[batch] = setup_dynamic_solver(
                self.geometry, self._energy, constrained_block_DOF_pairs=self.constrained_block_DOF_pairs,
                constrained_DOFs_fn=self._drive, damped_blocks=np.arange(self.geometry.n_blocks), rtol=self.rtol, atol=self.atol,
                integrator=self.integrator, batch=batch, device=self.device, streams=self.streams, _lib=self._lib)
        return sd

    def control_params(self, design, amplitude, loading_rate, compressive_strain, compressive_strain_rate):
        centroids, cnv = geometry_from_design_cached(self.geometry, design)
        return ControlParams(
            geometrical_params=GeometricalParams(block_centroids=centroids, centroid_node_vectors=cnv),
            mechanical_params=MechanicalParams(
                bond_params=LigamentParams(self.k_stretch, self.k_shear, self.k_rot, self.reference_bond_vectors),
                density=self.density, damping=self.damping,
                contact_params=ContactParams(min_angle=self.min_angle, cutoff_angle=self.cutoff_angle, k_contact=self.k_contact)),
            constraint_params=dict(amplitude=amplitude, loading_rate=loading_rate, compressive_strain=compressive_strain,
                                   compressive_strain_rate=compressive_strain_rate, input_delay=0.1 / loading_rate))       # :213

    def timepoints_of(self, loading_rate, compressive_strain, compressive_strain_rate, full_simulation_time=False, n_timepoints=None):
        """``:246-259``."""
        n = self.n_timepoints if n_timepoints is None else n_timepoints
        t0 = compressive_strain / compressive_strain_rate + 0.1 / loading_rate
        if full_simulation_time:
            return np.linspace(0, self.simulation_time_dynamic + t0, n)
        return np.concatenate([[0.], np.linspace(t0, t0 + self.simulation_time_dynamic, n)])

    def step_counts(self, timepoints, full_simulation_time):
        if self.steps_per_interval is None:
            return None
        counts = np.full(len(timepoints) - 1, int(self.steps_per_interval), dtype=np.int32)
        if not full_simulation_time:
            if self.static_steps is None:
                raise ValueError("steps_per_interval needs static_steps (RK steps of the static interval) as well")
            counts[0] = int(self.static_steps)
        return counts

    def solve_rows(self, design, rows, full_simulation_time=False, n_timepoints=None, keep_trajectory=False, want_fields=True,
                   group_key=None, after_group=None):
        """Integrate the forward inputs ``rows`` (R, 4) = (amplitude, loading_rate, compressive_strain, compressive_strain_rate) for
        one design.  Rows with identical output times (and identical ``group_key(r)`` when given) share an engine call.  Returns the
        list of SolutionData in row order (``want_fields``) and remembers the groups (``self.groups``: solver, row indices, params).
        ``after_group(solver, row indices, params)`` runs right after a group's solve, while its trajectory is still the engine's
        (groups of equal size share a solver: the next solve replaces it)."""
        rows = np.asarray(rows, dtype=float).reshape(-1, 4)
        tps = [self.timepoints_of(r[1], r[2], r[3], full_simulation_time, n_timepoints) for r in rows]
        # with explicit step counts every row may keep its own time grid inside one call (dfx_forward_grid_members: shared step
        # counts, per-member times); the adaptive controller chooses one grid per call, so there only equal grids share a call
        own_grids = self.steps_per_interval is not None
        keys = {}
        for i, tp in enumerate(tps):
            keys.setdefault((len(tp) if own_grids else tp.tobytes(), None if group_key is None else group_key(i)), []).append(i)
        sols, self.groups = [None] * len(rows), []
        for idx in keys.values():
            sd = self.solver(len(idx))
            cps = [self.control_params(design, *rows[i]) for i in idx]
            tp = tps[idx[0]]
            same = all(np.array_equal(tps[i], tp) for i in idx)
            tp_call = tp if same else np.stack([tps[i] for i in idx])
            fields = sd(self.state0, tp_call, cps if len(idx) > 1 else cps[0], keep_trajectory=keep_trajectory,
                        steps_per_interval=self.step_counts(tp, full_simulation_time), want_fields=want_fields)
            self.groups.append((sd, idx, cps, tp_call))
            if after_group is not None:
                after_group(sd, idx, cps)
            if fields is None:
                continue
            for m, i in enumerate(idx):
                f = fields[m] if len(idx) > 1 else fields
                gp = cps[m].geometrical_params
                tp = tps[i]
                sols[i] = SolutionData(gp.block_centroids, gp.centroid_node_vectors, self.bond_connectivity,
                                       tp if full_simulation_time else tp[1:] - tp[1], f if full_simulation_time else f[1:])    # :268-276
        self._last_design = design
        if want_fields:
            self._last_solutions, self.solution_data = sols, sols[0]
            self._last_solve_id = -1        # response data of a static-tuning solution: host formulas (rows may live in several engines)
        return sols if want_fields else None

    def solve(self, design, amplitude, loading_rate, compressive_strain, compressive_strain_rate, full_simulation_time=False,
              n_timepoints=None):
        """``forward(...)`` of ``:199-277`` for one forward input."""
        return self.solve_rows(design, [[amplitude, loading_rate, compressive_strain, compressive_strain_rate]],
                               full_simulation_time, n_timepoints)[0]

    def solve_dynamic(self, design, amplitude, loading_rate, compressive_strain, compressive_strain_rate):
        """``:280-281``: the dynamic step only (what the optimisation differentiates)."""
        return self.solve(design, amplitude, loading_rate, compressive_strain, compressive_strain_rate, False, self.n_timepoints)

    compute_response_data = _compute_response_data
    to_dict = _forward_to_dict
    from_dict = classmethod(_forward_from_dict)
    to_data = _forward_to_data
    from_data = classmethod(_forward_from_data)


class StaticTuningKineticEnergy:
    """``problems/quads_kinetic_energy_static_tuning.py:OptimizationProblem.setup_objective`` (``:430-480``):
    ``weights @ [target kinetic energy of every forward input]`` for ONE design; negative weights "protect".  Every row has its own
    target blocks (``target_sizes`` / ``target_shifts``).  ``value_and_grad`` also leaves the gradient w.r.t. the forward inputs in
    ``self.last_input_grads`` (one dict per row; ``loading_rate`` includes the path through ``input_delay = 0.1 / loading_rate``;
    output times are held fixed: the engine has no time-point cotangent and the reference's optimisation never asks for one)."""

    def __init__(self, forward, forward_input, target_sizes, target_shifts, weights):
        self.forward = forward
        if not getattr(forward, "is_setup", False):
            forward.setup()
        self.forward_input = forward_input
        self.rows = forward_input.rows()
        self.weights = np.asarray(weights, dtype=float)
        self.target_blocks = [quads_target_blocks(forward.geometry, ts, sh) for ts, sh in zip(target_sizes, target_shifts)]
        if not (len(self.rows) == len(self.weights) == len(self.target_blocks)):
            raise ValueError("forward inputs, weights and targets must have one entry per forward problem")
        # The device objective sums over every output row, the extra t = 0 row of the dynamic-step grid included; value() and the
        # reference drop that row (:268-276).  The two agree because a target block is at rest at t = 0 -- unless it carries a prescribed
        # DOF (the capped ramp gives the clamped rows a velocity L * rate / 2 from the first instant), whose kinetic energy the reverse
        # sweep would not differentiate w.r.t. the drive either.  The reference's targets are interior blocks: refuse the others.
        prescribed = np.unique(np.asarray(forward.constrained_block_DOF_pairs).reshape(-1, 2)[:, 0])
        for tb in self.target_blocks:
            if np.intersect1d(tb, prescribed).size:
                raise ValueError("StaticTuningKineticEnergy: a target block carries a prescribed DOF (clamped / driven rows); "
                                 "choose targets inside the lattice")

    def individual(self, design):
        fw = self.forward
        sols = fw.solve_rows(design, self.rows)
        return np.array([E.kinetic_energy(s.fields[:, 1, tb, :], compute_inertia(s.centroid_node_vectors, fw.density)[tb])
                         for s, tb in zip(sols, self.target_blocks)])

    def value(self, design):
        return float(self.weights @ self.individual(design))

    def value_and_grad(self, design):
        fw = self.forward
        vals = np.zeros(len(self.rows))
        sums, self.last_input_grads = {}, [None] * len(self.rows)

        def reverse(sd, idx, cps):
            obj, raw = sd.kinetic_energy_value_and_raw(self.target_blocks[idx[0]],
                                                       which=("centroid_node_vectors", "void_angle0", "inertia", "fn_params"))
            for m, i in enumerate(idx):
                vals[i] = obj[m]
                bar = {}
                for f, term in enumerate(sd.con_terms):
                    term.scatter_grad(raw["fn_params"][m][f], bar, cps[m].constraint_params)
                bar["loading_rate"] = bar.get("loading_rate", 0.0) - bar.pop("input_delay", 0.0) * 0.1 / self.rows[i][1] ** 2
                self.last_input_grads[i] = bar
                for k in raw:
                    if k != "fn_params":
                        sums[k] = sums.get(k, 0.0) + self.weights[i] * np.asarray(raw[k][m], dtype=float)

        fw.solve_rows(design, self.rows, keep_trajectory=True, want_fields=False,
                      group_key=lambda i: self.target_blocks[i].tobytes(), after_group=reverse)
        self.last_individual = vals
        grads = design_gradients(fw, [design], {k: a[None] for k, a in sums.items()})[0]
        return float(self.weights @ vals), grads


def design_gradients(fw, designs, raw):
    """The engine's raw gradients (batch-leading arrays for centroid_node_vectors, void_angle0, inertia[, block_centroids]) mapped
    back to the designs: void-angle and inertia chain rules, then the lattice map (all linear in the cotangent) -- one native pass over the
    blocks of all designs (``dfx_design_vjp``) where the native map applies, the NumPy maps of geometry.py otherwise."""
    from .geometry import compute_inertia_vjp, void_angles0_vjp
    geo, bonds = fw.geometry, fw.solve_dynamics.bonds
    nm = _native_design_map(fw)
    if nm is not None:
        lib, ndm, _ = nm
        va = raw.get("void_angle0")
        if va is not None and not np.any(va):
            va = None           # (contacts are rare: an identically zero cotangent maps to exact zeros)
        return ndm.vjp(lib, designs, float(fw.density), raw["centroid_node_vectors"], raw.get("block_centroids"), raw["inertia"], va)
    grads = []
    for m, d in enumerate(designs):
        _, cnv = geometry_from_design_cached(geo, d)
        cnv_bar = np.array(raw["centroid_node_vectors"][m], dtype=float)
        # (contacts are rare: an identically zero cotangent -- the engine hands out a view of a zero buffer then -- maps to exact zeros)
        if "void_angle0" in raw and np.any(raw["void_angle0"][m]):
            cnv_bar += void_angles0_vjp(cnv, bonds, raw["void_angle0"][m])
        # inertia = density * (area, area, polar moment about the centroid): its cotangent rides on the lattice map's own polygon pass
        ib = np.asarray(raw["inertia"][m], dtype=float)
        rho = np.broadcast_to(np.asarray(fw.density, dtype=float), ib.shape[:1])
        grads.append(geo.vjp(d, cnv_bar, raw["block_centroids"][m] if "block_centroids" in raw else None,
                             props_bar=(rho * (ib[:, 0] + ib[:, 1]), rho * ib[:, 2])))
    return grads


class TargetKineticEnergy:
    """objective(design) = sum_t sum_{b in target} m v^2/2 and its gradient w.r.t. the design
    (problems/quads_focusing.py:432-471 + jit(value_and_grad(.)) at :565; kagome_focusing.py:388-424)."""

    def __init__(self, forward, target_size, target_shift):
        self.forward = forward
        if not getattr(forward, "is_setup", False):
            forward.setup()
        pick = kagome_target_blocks if isinstance(forward.geometry, KagomeGeometry) else quads_target_blocks
        self.target_size, self.target_shift = tuple(target_size), tuple(target_shift)
        self.target_blocks = pick(forward.geometry, target_size, target_shift)

    def value(self, design):
        sol = self.forward.solve(design)
        sols = sol if isinstance(sol, list) else [sol]
        vals = [E.kinetic_energy(s.fields[:, 1, self.target_blocks, :],
                                 compute_inertia(s.centroid_node_vectors, self.forward.density)[self.target_blocks]) for s in sols]
        return vals if isinstance(sol, list) else vals[0]

    def value_and_grad(self, design):
        """Only the parameter groups a design reaches are differentiated on the device (node vectors, undeformed void angles,
        inertia, block centroids for the distance-based contact): asking the engine for every ControlParams leaf switches the
        reverse stage to its variant that also accumulates per-ligament stiffness / reference-vector / contact-constant / damping
        gradients -- twice the time per launch on the 64x64 kagome of config 4 (profiles/r02_c4_design_gradient_subset.txt)."""
        fw = self.forward
        if isinstance(design, list) and len(design) > fw.batch and len(design) % fw.batch == 0:
            # more designs than the engine integrates side by side: one call after the other (an ensemble whose checkpoint does not fit
            # the device at once -- config 4 as written, 64 designs x 75 000 steps on ONE GPU -- runs as two calls of 32)
            parts = [self.value_and_grad(design[i:i + fw.batch]) for i in range(0, len(design), fw.batch)]
            return np.concatenate([p[0] for p in parts]), [g for p in parts for g in p[1]]
        fw.solve(design, keep_trajectory=True, want_fields=False)
        obj, raw = fw.solve_dynamics.kinetic_energy_value_and_raw(self.target_blocks)
        # device time of this evaluation (forward + reverse sweep), for throughput reports
        self.device_ms_forward = getattr(self, "device_ms_forward", 0.0) + fw.solve_dynamics.stats["kernel_ms"]
        self.device_ms_adjoint = getattr(self, "device_ms_adjoint", 0.0) + fw.solve_dynamics.adjoint_stats["kernel_ms"]
        self.device_ms = getattr(self, "device_ms", 0.0) + fw.solve_dynamics.stats["kernel_ms"] + fw.solve_dynamics.adjoint_stats["kernel_ms"]
        many = isinstance(design, list)
        designs = design if many else [design]
        grads = design_gradients(fw, designs, raw)
        obj = np.asarray(obj, dtype=float)
        return (obj, grads) if many else (float(obj[0]), grads[0])


class SplitTargetKineticEnergy:
    """objective(design) = weights @ [kinetic energy of every target region] for ONE forward problem -- the energy of a single input
    split between several targets (problems/quads_energy_splitting.py:14-88).  One forward solve and ONE reverse sweep: the cotangent
    of the weighted sum goes through ``solve_dynamics.vjp`` (w_k m v on the velocities of target k), the explicit dependence on the
    inertia of the target blocks (w_k sum_t v^2 / 2) is added here."""

    def __init__(self, forward, target_sizes, target_shifts, weights):
        self.forward = forward
        if not getattr(forward, "is_setup", False):
            forward.setup()
        pick = kagome_target_blocks if isinstance(forward.geometry, KagomeGeometry) else quads_target_blocks
        self.target_sizes, self.target_shifts = tuple(map(tuple, target_sizes)), tuple(map(tuple, target_shifts))
        if len(self.target_sizes) != len(self.target_shifts) or len(weights) != len(self.target_sizes):
            raise ValueError("target_sizes, target_shifts and weights must have the same length")
        self.weights = np.asarray(weights, dtype=float)
        self.target_blocks_list = [pick(forward.geometry, ts, sh) for ts, sh in zip(self.target_sizes, self.target_shifts)]

    def _individual(self, sol):
        inertia = compute_inertia(sol.centroid_node_vectors, self.forward.density)
        return np.array([E.kinetic_energy(sol.fields[:, 1, tb, :], inertia[tb]) for tb in self.target_blocks_list]), inertia

    def individual(self, design):
        """quads_energy_splitting.py:66-83 (``objective_fn_individual``)."""
        return self._individual(self.forward.solve(design))[0]

    def value(self, design):
        return float(self.weights @ self.individual(design))

    def value_and_grad(self, design):
        fw = self.forward
        sol = fw.solve(design, keep_trajectory=True)
        vals, inertia = self._individual(sol)
        self.last_individual = vals
        fb = np.zeros_like(sol.fields)
        raw_m = np.zeros((fw.geometry.n_blocks, 3))
        for w, tb in zip(self.weights, self.target_blocks_list):
            v = sol.fields[:, 1, tb, :]
            fb[:, 1, tb, :] += w * inertia[tb] * v            # (overlapping targets add up)
            np.add.at(raw_m, tb, w * 0.5 * (v ** 2).sum(0))
        raw = {k: np.array(v, dtype=float) for k, v in fw.solve_dynamics.vjp_raw(fb).items()}
        raw["inertia"][0] += raw_m
        return float(self.weights @ vals), design_gradients(fw, [design], raw)[0]


class TargetAngularMomentum:
    """objective(design) = sum_t sum_{b in target} [ (c_b + u_b - p) x (m v_b) + J omega_b ]: angular momentum of the
    target blocks about ``spin_center`` summed over the output times (problems/quads_spin.py:380-430 with
    energy.py:502-519).  Evaluated on the host from the fields; its cotangent goes through ``solve_dynamics.vjp`` (any
    objective that is a function of the fields can be written like this), the explicit dependence on the design through
    block centroids and inertia is added here.  ``spin_center``: a point, or "center" = mean centroid of the target
    blocks in the design the objective was set up with (quads_spin.py:400-402)."""

    def __init__(self, forward, target_size, target_shift, spin_center="center", reference_design=None):
        self.forward = forward
        if not getattr(forward, "is_setup", False):
            forward.setup()
        pick = kagome_target_blocks if isinstance(forward.geometry, KagomeGeometry) else quads_target_blocks
        self.target_blocks = pick(forward.geometry, target_size, target_shift)
        if isinstance(spin_center, str):
            if reference_design is None:
                raise ValueError("spin_center='center' needs the reference design")
            spin_center = forward.geometry.block_centroids(*reference_design)[self.target_blocks].mean(0)
        self.spin_center = np.asarray(spin_center, dtype=float)

    def _value(self, sol):
        from .geometry import compute_inertia
        tb = self.target_blocks
        inertia = compute_inertia(sol.centroid_node_vectors, self.forward.density)[tb]
        pos = sol.block_centroids[tb][None] + sol.fields[:, 0, tb, :2] - self.spin_center    # (T, nt, 2)
        vel = sol.fields[:, 1, tb, :]
        lin = pos[..., 0] * vel[..., 1] * inertia[:, 1] - pos[..., 1] * vel[..., 0] * inertia[:, 0]
        return float(np.sum(lin + vel[..., 2] * inertia[:, 2])), inertia, pos, vel

    def value(self, design):
        return self._value(self.forward.solve(design))[0]

    def value_and_grad(self, design):
        from .geometry import compute_inertia_vjp
        fw, tb = self.forward, self.target_blocks
        sol = fw.solve(design, keep_trajectory=True)
        val, inertia, pos, vel = self._value(sol)
        fb = np.zeros_like(sol.fields)
        fb[:, 0, tb, 0] = vel[..., 1] * inertia[:, 1]
        fb[:, 0, tb, 1] = -vel[..., 0] * inertia[:, 0]
        fb[:, 1, tb, 0] = -pos[..., 1] * inertia[:, 0]
        fb[:, 1, tb, 1] = pos[..., 0] * inertia[:, 1]
        fb[:, 1, tb, 2] = inertia[:, 2]
        # only the parameter groups a design reaches are differentiated on the device (TargetKineticEnergy.value_and_grad)
        raw = {k: np.array(v, dtype=float) for k, v in fw.solve_dynamics.vjp_raw(fb).items()}
        # explicit terms: positions contain the block centroids; the inertia of the target blocks
        cen_bar = raw["block_centroids"] if "block_centroids" in raw else np.zeros((1, fw.geometry.n_blocks, 2))
        cen_bar[0, tb, 0] += (vel[..., 1] * inertia[:, 1]).sum(0)
        cen_bar[0, tb, 1] += (-vel[..., 0] * inertia[:, 0]).sum(0)
        raw["block_centroids"] = cen_bar
        raw["inertia"][0, tb, 0] += (-pos[..., 1] * vel[..., 0]).sum(0)
        raw["inertia"][0, tb, 1] += (pos[..., 0] * vel[..., 1]).sum(0)
        raw["inertia"][0, tb, 2] += vel[..., 2].sum(0)
        return val, design_gradients(fw, [design], raw)[0]


class MultiInputTargetKineticEnergy:
    """weights @ [target kinetic energy of every forward problem], all sharing one design
    (problems/quads_focusing_multi_input.py:43-86).  The forward problems differ in their boundary conditions
    (loaded side / input shift), so each owns a solver; the design gradient is the weighted sum."""

    def __init__(self, forward_problems, target_size, target_shift, weights):
        # Several engines can be driven at once (one host thread per input).  That pays while ONE engine does not fill the chip:
        # 24x16 x 96 members x 3 inputs, one stream per engine + three threads 0.91 s per evaluation against 1.4 s with two member
        # groups per engine run one after the other (profiles/r02_multi_engine_streams.txt).  With 256 members each engine fills the
        # chip alone and three threads issuing eager launches only contend for the runtime (forward 0.70 s vs 0.31 s per call,
        # profiles/r02_c5_host_profile.txt): those keep the engine's own choice and run the inputs one after the other.
        for fp in forward_problems:
            if len(forward_problems) > 1 and not fp.streams:
                n_units = (fp.n1_blocks * fp.n2_blocks) if hasattr(fp, "n1_blocks") else 2 * fp.n1_cells * fp.n2_cells
                if fp.batch * n_units * 4 // 64 < 4096:      # waves per launch: below two rounds of the chip
                    fp.streams = 1
                    if getattr(fp, "is_setup", False):       # already built with the engine's own choice: build it again
                        fp.solve_dynamics.engine.close()
                        fp.setup()
        self.objectives = [TargetKineticEnergy(fp, target_size, target_shift) for fp in forward_problems]
        self.target_size, self.target_shift = tuple(target_size), tuple(target_shift)
        self.concurrent_inputs = True
        # Inputs that run one after the other (forward + reverse sweep of one input before the next starts) keep their trajectory
        # checkpoints in ONE set of buffers: a third of the memory and of the allocation time, and room for the records level where
        # three separate checkpoints only fitted the stage accelerations (config 5: 256 designs x 3 inputs on one GPU).
        if len(forward_problems) > 1 and not all(fp.streams == 1 for fp in forward_problems) \
                and len({(fp.device, id(fp._lib)) for fp in forward_problems}) == 1:
            first = forward_problems[0].solve_dynamics.engine
            for fp in forward_problems[1:]:
                fp.solve_dynamics.engine.share_checkpoint(first)
            self.concurrent_inputs = False
        self.weights = np.asarray(weights, dtype=float)
        self.target_blocks = self.objectives[0].target_blocks
        self.forward = forward_problems[0]

    def individual(self, design):
        return np.array([o.value(design) for o in self.objectives])

    def value(self, design):
        return float(self.weights @ self.individual(design))

    def value_and_grad(self, design):
        """design: one design tuple, or a list of them (one per ensemble member of the batched forward problems).
        The engine's raw gradients of the individual inputs are summed with the weights first; the chain rules back to the
        design (void angles, inertia, lattice map -- all linear in the cotangent and all functions of the SAME design) are
        then applied once per design instead of once per input."""
        from .geometry import compute_inertia_vjp, void_angles0_vjp
        many = isinstance(design, list)
        designs = design if many else [design]
        def one(o):
            fw = o.forward
            fw.solve(design, keep_trajectory=True, want_fields=False)
            v, g = fw.solve_dynamics.kinetic_energy_value_and_raw(o.target_blocks)
            o.device_ms = getattr(o, "device_ms", 0.0) + fw.solve_dynamics.stats["kernel_ms"] + fw.solve_dynamics.adjoint_stats["kernel_ms"]
            return v, g

        # The inputs are independent solves on their own engine handles (own HIP streams); the C calls release the GIL, so
        # running them from a few threads lets their launches overlap on the GPU -- small lattices are launch-bound, one
        # input alone leaves most of the chip idle.  (Geometry of the designs first, once, from this thread: the cache.)
        for d in designs:
            geometry_from_design_cached(self.objectives[0].forward.geometry, d)
        # Only when every engine advances its members as ONE group on ONE stream (small ensembles; known from the statistics of
        # the previous solve, so the first evaluation runs the inputs one after the other): three engines with two member
        # groups each -- six streams with fork/join events -- were seen to stall on the four hardware queues.
        single_stream = all(getattr(o.forward.solve_dynamics, "stats", {}).get("streams", 0) == 1 for o in self.objectives)
        if self.concurrent_inputs and single_stream and len(self.objectives) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(len(self.objectives)) as pool:
                results = list(pool.map(one, self.objectives))
        else:
            results = [one(o) for o in self.objectives]
        vals, sums = [], None
        for w, (v, g) in zip(self.weights, results):
            vals.append(np.asarray(v, dtype=float))
            if sums is None:
                sums = {k: w * a for k, a in g.items()}
            else:
                for k, a in g.items():
                    sums[k] += w * a
        self.last_individual = np.array(vals) if many else np.array(vals)[:, 0]
        value = self.weights @ np.array(vals)
        grads = design_gradients(self.objectives[0].forward, designs, sums)
        if many:
            return list(value), grads
        return float(value[0]), grads[0]


def _ensemble_member(g, x0, n_iterations, lower_bound, upper_bound, min_void_angle, min_block_angle, min_edge_length):
    """One member of :func:`run_ensemble_optimization` as a coroutine (``yield x`` asks for the objective and its gradient at x):
    the reference's loop for ONE design.  A module-level function of picklable arguments, so that it can live in a host worker
    process (``optimize.MemberWorkers``); the constraint history travels back inside the result."""
    from .optimize import maximizing, mma_steps
    violation = {"angles": [], "edge_lengths": []}
    cons = []
    if min_void_angle is not None and min_block_angle is not None:
        def ca(x):
            r = angle_constraints(g, _unflatten_design(g, x), min_void_angle, min_block_angle)
            violation["angles"].append(float(r.max()))
            return r
        cons.append((ca, lambda x: angle_constraints_jac(g, _unflatten_design(g, x))))
    if min_edge_length is not None:
        def ce(x):
            r = edge_length_constraints(g, _unflatten_design(g, x), min_edge_length)
            violation["edge_lengths"].append(float(r.max()))
            return r
        cons.append((ce, lambda x: edge_length_constraints_jac(g, _unflatten_design(g, x))))
    res = yield from maximizing(mma_steps(x0, lower=lower_bound, upper=upper_bound, constraints=cons, maxeval=n_iterations,
                                          constraint_tol=1e-8))
    res["constraints_violation"] = violation
    return res


def run_ensemble_optimization(objective, initial_guesses, n_iterations, lower_bound=None, upper_bound=None,
                              min_void_angle=None, min_block_angle=None, min_edge_length=None, verbose=False, workers=None,
                              pipeline=False):
    """``len(initial_guesses)`` independent design optimisations (BASELINE config 5: an ensemble of multi-input focusing
    designs) advancing in lock-step: every round the pending design of EVERY member is evaluated in one batched call of
    the objective (``objective.value_and_grad(list of designs)`` -- the forward problems were built with
    ``batch=len(initial_guesses)``), i.e. one forward + reverse sweep of the engine integrates all members side by side.
    Each member runs the reference's loop (method of moving asymptotes under the angle / edge-length constraints,
    problems/quads_focusing.py:546-652) exactly as it would alone.  ``workers`` (``optimize.MemberWorkers``, created before the
    first GPU call): host processes that run the members' constraint evaluations and MMA sub-problems side by side.
    ``pipeline=True`` (needs ``workers``; the forward problems built with ``batch = len(initial_guesses) // 2``): the two halves of
    the ensemble take turns on the device, the host work of one half (MMA sub-problems, constraint Jacobians) runs while the other
    half is integrated -- same iterates, less wall time.
    Returns (best designs, list of per-member dicts with objective_values / constraints_violation / mma result)."""
    from .optimize import drive_ensemble
    g = objective.forward.geometry
    n = len(initial_guesses)
    logs = [dict(objective_values=[]) for _ in range(n)]
    if pipeline and getattr(objective.forward, "batch", n // 2) != n // 2:
        raise ValueError(f"pipeline=True evaluates half of the ensemble per engine call: build the forward problems with batch={n // 2}")

    round_times = []

    def batch_fun(xs, ids=None):
        import time
        t0 = time.perf_counter()
        designs = [_unflatten_design(g, x) for x in xs]
        vals, grads = objective.value_and_grad(designs)
        for k, v in enumerate(vals):
            i = k if ids is None else ids[k]                       # pipelined: one half of the ensemble per call
            if len(logs[i]["objective_values"]) < n_iterations:
                logs[i]["objective_values"].append(float(v))
        round_times.append(time.perf_counter() - t0)
        if verbose:
            print(f"round {len(round_times)}: evaluation of all members {round_times[-1]:.2f} s, objectives = "
                  f"{np.array2string(np.asarray(vals), precision=4, threshold=8)}")
        return [(float(v), _flatten_design(gr)) for v, gr in zip(vals, grads)]

    specs = [(_ensemble_member, (g, _flatten_design(d), n_iterations, lower_bound, upper_bound, min_void_angle, min_block_angle,
                                 min_edge_length), {}) for d in initial_guesses]
    res = drive_ensemble(batch_fun, specs, workers, groups=2 if pipeline else 1)
    for log, r in zip(logs, res):
        log["constraints_violation"] = r.pop("constraints_violation")
        log["mma"] = r
        log["evaluation_seconds"] = list(round_times)      # wall time of every lock-step evaluation of the whole ensemble (the first one
                                                           # includes the engines' device allocations)
    return [_unflatten_design(g, r.x) for r in res], logs


# -- geometric constraints of the optimisation (problems/quads_focusing.py:473-544) ------------------------------------

def quads_boundary_nodes(geometry):
    """Nodes of the rim blocks that face outwards and carry no ligament (problems/quads_focusing.py:477-489): bottom edge (node 3),
    right edge (node 0), top edge (node 1, right to left), left edge (node 2)."""
    n1, nb = geometry.n1_blocks, geometry.n_blocks
    return np.concatenate([np.arange(n1) * 4 + 3, np.arange(n1 - 1, nb, n1) * 4 + 0, np.arange(nb - 1, nb - n1 - 1, -1) * 4 + 1,
                           np.arange(0, nb, n1) * 4 + 2])


def _boundary_edges(geometry, ref):
    """Block, node, next and previous node, and the two edge vectors at the boundary nodes (geometry.py:181-202)."""
    nodes = quads_boundary_nodes(geometry)
    n = geometry.n_npb
    b, l = nodes // n, nodes % n
    nx, pv = (l + 1) % n, (l - 1) % n
    return b, l, nx, pv, ref[b, nx] - ref[b, l], ref[b, pv] - ref[b, l]


def angle_constraints(geometry, design, min_void_angle=0., min_block_angle=0., boundary_angle_constraint=False):
    """<= 0 when satisfied: -(angle mod 2pi - min) for the two void and the two block angles of every bond; with
    ``boundary_angle_constraint`` also for the block angle at every boundary node (problems/quads_focusing.py:473-532)."""
    from .geometry import compute_edge_angles
    cnv = geometry.centroid_node_vectors(*design)
    a = np.mod(np.stack(compute_edge_angles(cnv, geometry.bond_connectivity())), 2 * np.pi)
    out = [-(a[0] - min_void_angle), -(a[1] - min_void_angle), -(a[2] - min_block_angle), -(a[3] - min_block_angle)]
    if boundary_angle_constraint:
        _, _, _, _, e_next, e_prev = _boundary_edges(geometry, cnv)
        ang = np.arctan2(e_next[:, 0] * e_prev[:, 1] - e_next[:, 1] * e_prev[:, 0], (e_next * e_prev).sum(1))
        out.append(-(np.mod(ang, 2 * np.pi) - min_block_angle))
    return np.concatenate(out)


def edge_length_constraints(geometry, design, min_edge_length):
    """<= 0 when satisfied (problems/quads_focusing.py:535-544)."""
    from .geometry import compute_edge_lengths
    return -(compute_edge_lengths(geometry.centroid_node_vectors(*design)).reshape(-1) - min_edge_length)


def design_index_map(geometry):
    """(n_blocks, n_npb, 2) int: which entry of the flattened design tuple every reference node coordinate adds.
    The lattices place every node at ``constant + one design shift`` (geometry.py:607-952), so the map is read off by
    pushing the entry numbers through ``reference_node_vectors``."""
    shapes = geometry.design_shapes()
    sizes = [int(np.prod(sh)) for sh in shapes]
    ids = np.split(np.arange(sum(sizes), dtype=float), np.cumsum(sizes)[:-1])
    zeros = [np.zeros(sh) for sh in shapes]
    ref = geometry.reference_node_vectors(*[i.reshape(sh) for i, sh in zip(ids, shapes)]) - geometry.reference_node_vectors(*zeros)
    return np.rint(ref).astype(np.int64)


def _flatten_design(design):
    return np.concatenate([np.asarray(a, dtype=float).ravel() for a in design])


def _unflatten_design(geometry, x):
    shapes = geometry.design_shapes()
    sizes = [int(np.prod(sh)) for sh in shapes]
    return tuple(p.reshape(sh) for p, sh in zip(np.split(np.asarray(x, dtype=float), np.cumsum(sizes)[:-1]), shapes))


def _angle_rows(rows, cols, vals, row0, imap, bu, iu, lu, u, bw, iw, lw, w, sign):
    """Append the triplets of d angle(u, w) / d design for angles numbered row0.., u = r[bu, iu] - r[bu, lu] etc.
    angle = atan2(u x w, u . w): d/du = -perp(u)/|u|^2, d/dw = perp(w)/|w|^2, perp(a) = (-a_y, a_x)."""
    n = len(u)
    r = row0 + np.arange(n)
    du = -np.stack([-u[:, 1], u[:, 0]], 1) / (u ** 2).sum(1)[:, None] * sign
    dw = np.stack([-w[:, 1], w[:, 0]], 1) / (w ** 2).sum(1)[:, None] * sign
    for blk, node, d in ((bu, iu, du), (bu, lu, -du), (bw, iw, dw), (bw, lw, -dw)):
        for c in range(2):
            rows.append(r); cols.append(imap[blk, node, c]); vals.append(d[:, c])


def angle_constraints_jac(geometry, design, boundary_angle_constraint=False):
    """Sparse Jacobian (4 n_bonds [+ n_boundary_nodes], n_design) of :func:`angle_constraints` (the reference takes ``jax.jacobian``
    of it, problems/quads_focusing.py:586-587).  Edge vectors are differences of node vectors of one block, so the centroid
    shift cancels and the reference node vectors can be differentiated directly."""
    import scipy.sparse as sp
    ref = geometry.reference_node_vectors(*design)
    bonds = np.asarray(geometry.bond_connectivity(), dtype=np.int64)
    n = geometry.n_npb
    b1, l1 = bonds[:, 0] // n, bonds[:, 0] % n
    b2, l2 = bonds[:, 1] // n, bonds[:, 1] % n
    n1, p1, n2, p2 = (l1 + 1) % n, (l1 - 1) % n, (l2 + 1) % n, (l2 - 1) % n
    e1p, e1m = ref[b1, n1] - ref[b1, l1], ref[b1, p1] - ref[b1, l1]
    e2p, e2m = ref[b2, n2] - ref[b2, l2], ref[b2, p2] - ref[b2, l2]
    imap = design_index_map(geometry)
    nb = len(bonds)
    rows, cols, vals = [], [], []
    # rows: void_1 = angle(e2m, e1p), void_2 = angle(e1m, e2p), block_1 = angle(e1p, e1m), block_2 = angle(e2p, e2m); constraint = -(angle - min)
    _angle_rows(rows, cols, vals, 0 * nb, imap, b2, p2, l2, e2m, b1, n1, l1, e1p, -1.0)
    _angle_rows(rows, cols, vals, 1 * nb, imap, b1, p1, l1, e1m, b2, n2, l2, e2p, -1.0)
    _angle_rows(rows, cols, vals, 2 * nb, imap, b1, n1, l1, e1p, b1, p1, l1, e1m, -1.0)
    _angle_rows(rows, cols, vals, 3 * nb, imap, b2, n2, l2, e2p, b2, p2, l2, e2m, -1.0)
    n_rows = 4 * nb
    if boundary_angle_constraint:
        bb, bl, bnx, bpv, e_next, e_prev = _boundary_edges(geometry, ref)
        _angle_rows(rows, cols, vals, n_rows, imap, bb, bnx, bl, e_next, bb, bpv, bl, e_prev, -1.0)
        n_rows += len(bb)
    n_design = int(sum(np.prod(sh) for sh in geometry.design_shapes()))
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_rows, n_design)).tocsr()


def edge_length_constraints_jac(geometry, design):
    """Sparse Jacobian (n_nodes, n_design) of :func:`edge_length_constraints`."""
    import scipy.sparse as sp
    ref = geometry.reference_node_vectors(*design)
    e = np.roll(ref, 1, axis=1) - ref                       # edge k = r[k-1] - r[k]
    unit = e / np.linalg.norm(e, axis=2, keepdims=True)
    imap = design_index_map(geometry)
    nbk, npb = ref.shape[:2]
    r = np.arange(nbk * npb).reshape(nbk, npb)
    rows, cols, vals = [], [], []
    for c in range(2):
        rows += [r.ravel(), r.ravel()]
        cols += [np.roll(imap, 1, axis=1)[:, :, c].ravel(), imap[:, :, c].ravel()]
        vals += [-unit[:, :, c].ravel(), unit[:, :, c].ravel()]   # constraint = -(length - min)
    n_design = int(sum(np.prod(sh) for sh in geometry.design_shapes()))
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nbk * npb, n_design)).tocsr()


class RestrictedDesignSpace:
    """Only the shifts inside a window of ``design_patch_size`` blocks centred on the target are design variables, the others stay at
    ``initial_guess_all`` (problems/quads_focusing_restricted_space.py:417-469): the masks, ``all_to_reduced_shifts`` and
    ``reduced_to_all_shifts`` of the reference, plus what an optimiser on the reduced vector needs -- the positions of the reduced
    variables in the flattened full design (gradient entries / Jacobian columns to keep)."""

    def __init__(self, n1_blocks, n2_blocks, initial_guess_all, target_shift, design_patch_size=None):
        self.initial_guess_all = tuple(np.array(a, dtype=float) for a in initial_guess_all)
        patch = design_patch_size if design_patch_size is not None else max(n1_blocks, n2_blocks)
        self.design_patch_size = patch
        x0 = int(np.clip((n1_blocks - patch) // 2 + target_shift[0], 0, n1_blocks))
        x1 = int(np.clip((n1_blocks + patch) // 2 + target_shift[0], 0, n1_blocks))
        y0 = int(np.clip((n2_blocks - patch) // 2 + target_shift[1], 0, n2_blocks))
        y1 = int(np.clip((n2_blocks + patch) // 2 + target_shift[1], 0, n2_blocks))
        self.horizontal_shifts_mask = np.full(self.initial_guess_all[0].shape, False)
        self.horizontal_shifts_mask[x0:x1 + 1, y0:y1] = True
        self.vertical_shifts_mask = np.full(self.initial_guess_all[1].shape, False)
        self.vertical_shifts_mask[x0:x1, y0:y1 + 1] = True
        self.columns = np.flatnonzero(np.concatenate([self.horizontal_shifts_mask.ravel(), self.vertical_shifts_mask.ravel()]))
        self.sizes = (int(self.horizontal_shifts_mask.sum()), int(self.vertical_shifts_mask.sum()))

    def all_to_reduced_shifts(self, all_shifts):
        horizontal_shifts, vertical_shifts = all_shifts
        return np.asarray(horizontal_shifts)[self.horizontal_shifts_mask], np.asarray(vertical_shifts)[self.vertical_shifts_mask]

    def reduced_to_all_shifts(self, reduced_shifts):
        horizontal_shifts, vertical_shifts = (a.copy() for a in self.initial_guess_all)
        horizontal_shifts[self.horizontal_shifts_mask] = reduced_shifts[0]
        vertical_shifts[self.vertical_shifts_mask] = reduced_shifts[1]
        return horizontal_shifts, vertical_shifts

    def unflatten(self, x):
        x = np.asarray(x, dtype=float)
        return x[:self.sizes[0]], x[self.sizes[0]:]


@dataclass
class OptimizationProblem:
    """Inverse design loop with the bookkeeping of the reference's ``OptimizationProblem``
    (problems/quads_focusing.py:408-690: objective_values, design_values, constraints_violation, to_dict).
    NLopt is not available on the target image, so ``run_optimization`` is a bound-projected gradient ASCENT with
    back-tracking that only accepts feasible designs (angle / edge-length constraints) -- the reference maximises the
    same objective with NLopt's LD_MMA."""
    objective: Any
    objective_values: Optional[list] = None
    design_values: Optional[list] = None
    constraints_violation: Optional[dict] = None
    name: str = "quads_focusing"
    # restricted design space (problems/quads_focusing_restricted_space.py:417-420): with ``initial_guess_all`` given, the loop of
    # ``run_optimization_nlopt`` runs on the reduced shifts (its ``initial_guess`` and ``design_values`` are reduced shifts, as in
    # the reference) and everything outside the patch stays at ``initial_guess_all``
    initial_guess_all: Optional[tuple] = None
    design_patch_size: Optional[int] = None
    # per-target / per-input values of every evaluation, for objectives that are weighted sums (quads_energy_splitting.py:26,
    # quads_kinetic_energy_static_tuning.py, quads_focusing_multi_input.py: ``objective_values_individual``)
    objective_values_individual: Optional[list] = None

    def __post_init__(self):
        self.objective_values_individual = [] if self.objective_values_individual is None else self.objective_values_individual
        self.objective_values = [] if self.objective_values is None else self.objective_values
        self.design_values = [] if self.design_values is None else self.design_values
        self.constraints_violation = {"angles": [], "edge_lengths": []} if self.constraints_violation is None else self.constraints_violation
        self.space = None
        if self.initial_guess_all is not None:
            g = self.objective.forward.geometry
            self.space = RestrictedDesignSpace(g.n1_blocks, g.n2_blocks, self.initial_guess_all, self.objective.target_shift,
                                               self.design_patch_size)
            self.design_patch_size = self.space.design_patch_size

    def all_to_reduced_shifts(self, all_shifts):
        return self.space.all_to_reduced_shifts(all_shifts)

    def reduced_to_all_shifts(self, reduced_shifts):
        return self.space.reduced_to_all_shifts(reduced_shifts)

    def violation(self, design, min_void_angle, min_block_angle, min_edge_length):
        g = self.objective.forward.geometry
        va = angle_constraints(g, design, min_void_angle, min_block_angle).max() if min_void_angle is not None and min_block_angle is not None else -np.inf
        ve = edge_length_constraints(g, design, min_edge_length).max() if min_edge_length is not None else -np.inf
        return va, ve

    def run_optimization(self, initial_guess, n_iterations, lower_bound=None, upper_bound=None, min_void_angle=None,
                         min_block_angle=None, min_edge_length=None, initial_step=None, verbose=True):
        x = tuple(np.array(a, dtype=float) for a in initial_guess)
        clip = (lambda d: tuple(np.clip(a, lower_bound, upper_bound) for a in d)) if (lower_bound is not None or upper_bound is not None) else (lambda d: d)
        v, g = self.objective.value_and_grad(x)
        scale = max(np.abs(a).max() for a in g) or 1.0
        step = initial_step if initial_step is not None else 0.01 * getattr(self.objective.forward.geometry, "spacing", 1.0)
        for it in range(n_iterations):
            self.objective_values.append(float(v))
            self.design_values.append(x)
            va, ve = self.violation(x, min_void_angle, min_block_angle, min_edge_length)
            self.constraints_violation["angles"].append(va)
            self.constraints_violation["edge_lengths"].append(ve)
            if verbose:
                print(f"Iteration: {len(self.objective_values)}\nObjective = {self.objective_values[-1]}")
            accepted = False
            for _ in range(12):
                trial = clip(tuple(a + step * ga / scale for a, ga in zip(x, g)))
                va, ve = self.violation(trial, min_void_angle, min_block_angle, min_edge_length)
                if max(va, ve) <= 1e-8:
                    vt, gt = self.objective.value_and_grad(trial)
                    if vt > v:
                        x, v, g, accepted = trial, vt, gt, True
                        scale = max(np.abs(a).max() for a in g) or 1.0
                        step *= 1.5
                        break
                step *= 0.5
            if not accepted:
                break
        self.objective_values.append(float(v))
        self.design_values.append(x)
        return x

    def run_optimization_nlopt(self, initial_guess, n_iterations, max_time=None, lower_bound=None, upper_bound=None,
                               min_void_angle=None, min_block_angle=None, min_edge_length=None, boundary_angle_constraint=False,
                               verbose=True):
        """The reference's loop (problems/quads_focusing.py:546-652) with the same arguments and bookkeeping: maximise the
        objective with the method of moving asymptotes (``difflexmm_amd.optimize``, standing in for ``nlopt.LD_MMA``)
        under the angle and edge-length inequality constraints, ``n_iterations`` objective evaluations at most."""
        import time
        from .optimize import mma_maximize
        g = self.objective.forward.geometry
        t0 = time.perf_counter()

        class _TimeUp(Exception):
            pass

        space = self.space
        # optimisation vector -> (what the histories keep, the full design the solver gets); gradients / Jacobians back
        if space is None:
            def designs_of(x):
                d = _unflatten_design(g, x)
                return d, d
            cols = None
        else:
            def designs_of(x):
                r = space.unflatten(x)
                return r, space.reduced_to_all_shifts(r)
            cols = space.columns

        def fun(x):
            if max_time is not None and self.objective_values and time.perf_counter() - t0 > max_time:
                raise _TimeUp
            kept, design = designs_of(x)
            v, grad = self.objective.value_and_grad(design)
            self.objective_values.append(float(v))
            if getattr(self.objective, "last_individual", None) is not None:
                self.objective_values_individual.append(np.array(self.objective.last_individual))
            self.design_values.append(kept)
            if verbose:
                print(f"Iteration: {len(self.objective_values)}\nObjective = {self.objective_values[-1]}")
            gflat = _flatten_design(grad)
            return float(v), gflat if cols is None else gflat[cols]

        def restrict(J):
            return J if cols is None else J[:, cols]

        constraints = []
        if min_void_angle is not None and min_block_angle is not None:
            def ca(x):
                r = angle_constraints(g, designs_of(x)[1], min_void_angle, min_block_angle, boundary_angle_constraint)
                self.constraints_violation["angles"].append(float(r.max()))
                return r
            constraints.append((ca, lambda x: restrict(angle_constraints_jac(g, designs_of(x)[1], boundary_angle_constraint))))
        if min_edge_length is not None:
            def ce(x):
                r = edge_length_constraints(g, designs_of(x)[1], min_edge_length)
                self.constraints_violation["edge_lengths"].append(float(r.max()))
                return r
            constraints.append((ce, lambda x: restrict(edge_length_constraints_jac(g, designs_of(x)[1]))))
        try:
            res = mma_maximize(fun, _flatten_design(initial_guess), lower=lower_bound, upper=upper_bound,
                               constraints=constraints, maxeval=n_iterations,
                               constraint_tol=1e-8)      # the reference passes 1e-8 per constraint to add_inequality_mconstraint
            self.mma_result = res
            best = designs_of(res.x)[0]
        except _TimeUp:
            best = self.design_values[int(np.argmax(self.objective_values))]
        return best

    def compute_best_forward(self):
        """problems/quads_focusing_restricted_space.py:649-660 (quads_focusing.py:654-664): forward solution of the last design."""
        if len(self.design_values) == 0:
            raise ValueError("No design has been optimized yet.")
        last = self.design_values[-1]
        last = self.space.reduced_to_all_shifts(last) if self.space is not None else last
        if hasattr(self.objective, "objectives"):       # several inputs share the design (quads_focusing_multi_input.py:230-244)
            return [o.forward.solve(last) for o in self.objective.objectives]
        if hasattr(self.objective, "forward_input"):
            return self.compute_best_forwards()
        return self.objective.forward.solve(last)

    def compute_best_forwards(self, n_timepoints: int = 200):
        """problems/quads_kinetic_energy_static_tuning.py:627-652: the last design solved for every forward input, dynamic step only,
        ``n_timepoints`` output times."""
        if len(self.design_values) == 0:
            raise ValueError("No design has been optimized yet.")
        fw = self.objective.forward
        fw.solution_data = [fw.solve(self.design_values[-1], *row, full_simulation_time=False, n_timepoints=n_timepoints)
                            for row in self.objective.forward_input.rows()]
        return fw.solution_data

    def to_dict(self):
        """problems/quads_focusing.py:686-690 (multi-input: quads_focusing_multi_input.py:183-189): the forward problem(s) as
        dicts + target placement + the histories of the loop, ready for ``utils.save_data``."""
        obj = self.objective
        out = dict(name=self.name, target_size=getattr(obj, "target_size", None), target_shift=getattr(obj, "target_shift", None),
                   objective_values=list(self.objective_values), design_values=list(self.design_values),
                   constraints_violation={k: list(v) for k, v in self.constraints_violation.items()})
        if self.space is not None:
            out["initial_guess_all"] = tuple(np.array(a) for a in self.space.initial_guess_all)
            out["design_patch_size"] = self.design_patch_size
        if self.objective_values_individual:
            out["objective_values_individual"] = [np.array(a) for a in self.objective_values_individual]
        if hasattr(obj, "target_sizes") and not hasattr(obj, "forward_input"):        # energy splitting (quads_energy_splitting.py:20-23)
            out.update(target_sizes=obj.target_sizes, target_shifts=obj.target_shifts, weights=[float(w) for w in obj.weights])
        if hasattr(obj, "objectives"):
            out["forward_problems"] = [o.forward.to_dict() for o in obj.objectives]
            out["weights"] = [float(w) for w in obj.weights]
        else:
            out["forward_problem"] = obj.forward.to_dict()
        return out

    def to_data(self):
        """problems/quads_focusing.py:671-672."""
        return OptimizationProblem.from_dict(self.to_dict(), _lib=getattr(self.objective.forward, "_lib", None)
                                             if hasattr(self.objective, "forward") else None)

    @staticmethod
    def from_data(optimization_data, _lib=None):
        """problems/quads_focusing.py:664-669."""
        return OptimizationProblem.from_dict(optimization_data if isinstance(optimization_data, dict) else optimization_data.to_dict(), _lib=_lib)

    @staticmethod
    def from_dict(dict_in, _lib=None):
        """problems/quads_focusing.py:677-684: rebuild the optimisation problem from a saved dict with its histories, so that a
        run can be inspected or RESUMED -- ``opt.run_optimization_nlopt(opt.design_values[-1], n_more, ...)`` appends to them."""
        d = dict(dict_in)

        def forward(fd):
            cls = KagomeFocusingForward if "n1_cells" in fd else QuadsFocusingForward
            return cls.from_dict(fd, _lib=_lib)
        if "target_sizes" in d and "forward_problem" in d:
            objective = SplitTargetKineticEnergy(forward(d["forward_problem"]), d["target_sizes"], d["target_shifts"], d["weights"])
        elif "forward_problems" in d:
            objective = MultiInputTargetKineticEnergy([forward(fd) for fd in d["forward_problems"]], d["target_size"], d["target_shift"],
                                                      d["weights"])
        else:
            objective = TargetKineticEnergy(forward(d["forward_problem"]), d["target_size"], d["target_shift"])
        return OptimizationProblem(objective, objective_values=list(d.get("objective_values", [])),
                                   design_values=list(d.get("design_values", [])),
                                   constraints_violation={k: list(v) for k, v in d.get("constraints_violation", {"angles": [], "edge_lengths": []}).items()},
                                   name=d.get("name", "quads_focusing"), initial_guess_all=d.get("initial_guess_all"),
                                   design_patch_size=d.get("design_patch_size"),
                                   objective_values_individual=list(d.get("objective_values_individual", [])))
