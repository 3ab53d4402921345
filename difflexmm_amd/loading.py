"""Time functions for prescribed displacements (``constrained_DOFs_fn``) and external forces (``loading_fn``).

The reference takes arbitrary Python callables here (``kinematics.py:40-81``, ``loading.py:12-47``); a HIP
kernel cannot call Python, so the engine supports the closed library of time functions that the reference's
``problems/``, ``scripts/`` and ``tests/`` actually use.  An object of this module is still a
callable with the reference's signature -- ``fn(t, **constraint_params)`` or ``fn(state, t, **loading_params)`` --
so host code can evaluate it, and it carries the declarative ``spec`` the engine consumes.

Every parameter is either a number (a constant baked into the solver) or a string naming the entry of
``control_params.constraint_params`` / ``loading_params`` that holds it (then it is differentiable).
"""
import numpy as np

from ._binding import (DFX_FN_PARAMS, FN_CONSTANT, FN_HARMONIC, FN_PULSE, FN_RAMP, FN_RAMP_CAP,
                       FN_SECH2TANH, FN_TABLE, FN_ZERO)


class TimeFunction:
    """vector * g(t; params).  ``vector`` is a scalar or one coefficient per constrained / loaded DOF."""
    type_id = FN_ZERO
    param_names = ()

    def __init__(self, vector=1.0, **params):
        self.vector = np.asarray(vector, dtype=float)
        unknown = set(params) - set(self.param_names)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unknown parameters {sorted(unknown)}")
        # default: look the parameter up under its own name
        self.params = {n: params.get(n, n) for n in self.param_names}

    def resolve(self, params_dict):
        """5 numbers for the engine from a constraint_params / loading_params dict."""
        out = np.zeros(DFX_FN_PARAMS)
        for i, n in enumerate(self.param_names):
            v = self.params[n]
            if isinstance(v, str):
                if v not in params_dict:
                    raise KeyError(f"{type(self).__name__}: parameter '{v}' missing from the params dict")
                v = params_dict[v]
            out[i] = float(v)
        return out

    def scatter_grad(self, grad5, out_dict, params_dict=None):
        """Add d/d(params) (5 numbers from the engine) to a dict keyed like the params dict (``params_dict``: the values the
        solve was run with, for terms whose engine parameters are derived from several entries)."""
        for i, n in enumerate(self.param_names):
            v = self.params[n]
            if isinstance(v, str):
                out_dict[v] = out_dict.get(v, 0.0) + float(grad5[i])

    def value(self, t, p):
        return 0.0

    def rate(self, t, p, eps=1e-7):
        """d value / dt (central difference in the scale of t; the kernels use the analytic form)."""
        h = eps * max(1.0, abs(t))
        return (self.value(t + h, p) - self.value(t - h, p)) / (2 * h)

    def param_partials(self, t, p, of="value"):
        """d value/dp or d rate/dp (5 numbers) by central differences -- only used on the host for the cotangents of
        prescribed-DOF OUTPUTS, a handful of numbers per solve."""
        f = self.value if of == "value" else self.rate
        out = np.zeros(DFX_FN_PARAMS)
        for i in range(len(self.param_names)):
            h = 1e-6 * max(1.0, abs(p[i]))
            pp, pm = p.copy(), p.copy()
            pp[i] += h; pm[i] -= h
            out[i] = (f(t, pp) - f(t, pm)) / (2 * h)
        return out

    def _eval(self, t, kwargs):
        return self.vector * self.value(float(t), self.resolve(kwargs))

    def __call__(self, *args, **kwargs):
        # fn(t, **params)  or  fn(state, t, **params)
        t = args[-1]
        return self._eval(t, kwargs)

    @property
    def terms(self):
        return [self]

    def __add__(self, other):
        return SumOfTimeFunctions(self.terms + other.terms)


class SumOfTimeFunctions(TimeFunction):
    def __init__(self, terms):
        self._terms = list(terms)

    @property
    def terms(self):
        return self._terms

    def _eval(self, t, kwargs):
        return sum(f._eval(t, kwargs) for f in self._terms)


class Zero(TimeFunction):
    """The reference's default ``lambda t: 0`` (dynamics.py:66)."""

    @property
    def terms(self):
        return []


class Pulse(TimeFunction):
    """amplitude/2 (1 - cos 2 pi f tau) on 0 < tau < 1/f, tau = t - input_delay (problems/quads_focusing.py:211-222)."""
    type_id = FN_PULSE
    param_names = ("amplitude", "loading_rate", "input_delay")

    def value(self, t, p):
        tau = t - p[2]
        return p[0] * 0.5 * (1 - np.cos(2 * np.pi * p[1] * tau)) if (tau > 0 and tau * p[1] < 1) else 0.0


class Harmonic(Pulse):
    """Same wave switched on at tau > 0 and never off (problems/quads_spin.py:210-222)."""
    type_id = FN_HARMONIC

    def value(self, t, p):
        tau = t - p[2]
        return p[0] * 0.5 * (1 - np.cos(2 * np.pi * p[1] * tau)) if tau > 0 else 0.0


class Ramp(TimeFunction):
    """amplitude * min(t * rate, 1)  (tests/test_difflexmm.py:85-86, problems/hinge_characterization.py:134-139)."""
    type_id = FN_RAMP
    param_names = ("amplitude", "rate")

    def value(self, t, p):
        return p[0] * (t * p[1] if t * p[1] < 1 else 1.0)


class Sech2Tanh(TimeFunction):
    """2A/s^2 sech^2(t/s - 3) tanh(3 - t/s)  (scripts/pulse_RS.py:49-50)."""
    type_id = FN_SECH2TANH
    param_names = ("amplitude", "width")

    def value(self, t, p):
        z = t / p[1] - 3.0
        return -2 * p[0] / p[1] ** 2 * (1 - np.tanh(z) ** 2) * np.tanh(z)


class Constant(TimeFunction):
    type_id = FN_CONSTANT
    param_names = ("amplitude",)

    def value(self, t, p):
        return p[0]


class CappedRamp(TimeFunction):
    """length * min(t * rate, cap): the static compression of ``problems/quads_kinetic_energy_static_tuning.py:176-182``
    (``length = (n2_blocks - 1) * spacing``, ``rate = compressive_strain_rate``, ``cap = compressive_strain``; the reference's
    ``jnp.where(t < cap / rate, t * rate, cap)``)."""
    type_id = FN_RAMP_CAP
    param_names = ("length", "rate", "cap")

    def value(self, t, p):
        return p[0] * (t * p[1] if t * p[1] < p[2] else p[2])


class DelayedPulse(Pulse):
    """The pulse of ``quads_kinetic_energy_static_tuning.py:184-186``: it starts once the static ramp has ended,
    ``tau = t - strain / strain_rate - input_delay``.  The engine integrates a plain pulse with the derived delay
    ``strain / strain_rate + input_delay``; its gradient w.r.t. that delay is chained back to all three on the host."""

    def __init__(self, vector=1.0, strain="compressive_strain", strain_rate="compressive_strain_rate", **params):
        super().__init__(vector, **params)
        self.strain, self.strain_rate = strain, strain_rate

    def _lookup(self, v, params_dict):
        if isinstance(v, str):
            if v not in params_dict:
                raise KeyError(f"{type(self).__name__}: parameter '{v}' missing from the params dict")
            return float(params_dict[v])
        return float(v)

    def resolve(self, params_dict):
        out = super().resolve(params_dict)
        out[2] += self._lookup(self.strain, params_dict) / self._lookup(self.strain_rate, params_dict)
        return out

    def scatter_grad(self, grad5, out_dict, params_dict=None):
        super().scatter_grad(grad5, out_dict, params_dict)
        if params_dict is None:
            raise ValueError("DelayedPulse.scatter_grad needs the params dict the solve was run with")
        eps, rate = self._lookup(self.strain, params_dict), self._lookup(self.strain_rate, params_dict)
        if isinstance(self.strain, str):
            out_dict[self.strain] = out_dict.get(self.strain, 0.0) + float(grad5[2]) / rate
        if isinstance(self.strain_rate, str):
            out_dict[self.strain_rate] = out_dict.get(self.strain_rate, 0.0) - float(grad5[2]) * eps / rate ** 2


def static_tuning_drive(static_vector, dynamic_vector, length):
    """``constrained_DOFs_fn`` of ``quads_kinetic_energy_static_tuning.py:176-186``: the static compression on
    ``static_vector`` plus the pulse, delayed by the end of the ramp + ``input_delay``, on ``dynamic_vector``; parameters
    ``amplitude, loading_rate, compressive_strain, compressive_strain_rate, input_delay`` of ``constraint_params``."""
    return (CappedRamp(static_vector, length=float(length), rate="compressive_strain_rate", cap="compressive_strain")
            + DelayedPulse(dynamic_vector))


class Table(TimeFunction):
    """amplitude * interp(t - delay; times, values): a recorded input signal, piecewise linear, end values held
    (``jnp.interp`` semantics; the reference resamples its experimental signals with it, problems/hinge_characterization.py:546-551).
    The table is static data of the solver; ``amplitude`` and ``delay`` are parameters."""
    type_id = FN_TABLE
    param_names = ("amplitude", "delay")

    def __init__(self, times, values, vector=1.0, amplitude=1.0, delay=0.0):
        super().__init__(vector, amplitude=amplitude, delay=delay)
        self.times, self.values = np.asarray(times, dtype=float), np.asarray(values, dtype=float)
        if self.times.ndim != 1 or self.times.shape != self.values.shape or len(self.times) < 2 or np.any(np.diff(self.times) <= 0):
            raise ValueError("Table: times must be 1-D, strictly increasing, as long as values, with >= 2 entries")
        self.table = (self.times, self.values)

    def value(self, t, p):
        return p[0] * float(np.interp(t - p[1], self.times, self.values))


zero = Zero()


def as_time_function(fn, what):
    """Accept the library objects (and None); anything else cannot run on the device."""
    if fn is None:
        return zero
    if isinstance(fn, TimeFunction):
        return fn
    raise TypeError(
        f"{what} must be built from difflexmm_amd.loading (Pulse, Harmonic, Ramp, Sech2Tanh, Constant, CappedRamp, DelayedPulse, Table, "
        f"zero, or a sum of them): an arbitrary Python callable cannot be evaluated inside a HIP kernel")
