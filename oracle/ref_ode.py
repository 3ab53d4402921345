"""Oracle restatement of ``jax.experimental.ode.odeint`` (jax 0.4.8) in NumPy (test infrastructure).

The integrator the reference calls at ``difflexmm/dynamics.py:166`` is third-party code that is
NOT under ``/root/reference`` (pinned ``jax 0.4.8`` / ``jaxlib 0.4.7``, ``poetry.lock:614-615,
664-665``).  This file restates its published algorithm: adaptive Dormand-Prince 5(4) with FSAL,
RMS error norm, the (0.9, 10, 0.2, order 5) step controller applied on accept and reject,
Hairer's initial step with order 4, and quartic dense output evaluated at the requested times
(steps are not clipped to output times).  It also provides the fixed-step variant of the same
tableau that the MI355X engine runs, so both can be compared on identical inputs.
"""
import numpy as np

# Dormand-Prince tableau (same numbers as jax.experimental.ode.runge_kutta_step)
ALPHA = np.array([1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0, 0.0])
BETA = np.array([
    [1 / 5, 0, 0, 0, 0, 0, 0],
    [3 / 40, 9 / 40, 0, 0, 0, 0, 0],
    [44 / 45, -56 / 15, 32 / 9, 0, 0, 0, 0],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729, 0, 0, 0],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656, 0, 0],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0],
])
C_SOL = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0])
C_ERR = np.array([35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
                  -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0])
C_MID = np.array([6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2,
                  -2691868925 / 45128329728 / 2, 187940372067 / 1594534317056 / 2,
                  -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2])


def runge_kutta_step(func, y0, f0, t0, dt):
    """One Dopri5 step: 6 new RHS evaluations, k[0] = f0 (FSAL), returns y1, f1=k[6], error, k."""
    k = np.zeros((7, y0.shape[0]))
    k[0] = f0
    for i in range(1, 7):
        ti = t0 + dt * ALPHA[i - 1]
        yi = y0 + dt * (BETA[i - 1] @ k)
        k[i] = func(yi, ti)
    y1 = dt * (C_SOL @ k) + y0
    return y1, k[6], dt * (C_ERR @ k), k


def fit_4th_order_polynomial(y0, y1, y_mid, dy0, dy1, dt):
    a = -2.0 * dt * dy0 + 2.0 * dt * dy1 - 8.0 * y0 - 8.0 * y1 + 16.0 * y_mid
    b = 5.0 * dt * dy0 - 3.0 * dt * dy1 + 18.0 * y0 + 14.0 * y1 - 32.0 * y_mid
    c = -4.0 * dt * dy0 + dt * dy1 - 11.0 * y0 - 5.0 * y1 + 16.0 * y_mid
    return np.stack([a, b, c, dt * dy0, y0])


def interp_fit_dopri(y0, y1, k, dt):
    y_mid = y0 + dt * (C_MID @ k)
    return fit_4th_order_polynomial(y0, y1, y_mid, k[0], k[6], dt)


def initial_step_size(func, t0, y0, order, rtol, atol, f0):
    """Hairer-Norsett-Wanner II.4 as restated by jax (order=4 at the call site)."""
    scale = atol + np.abs(y0) * rtol
    d0 = np.linalg.norm(y0 / scale)
    d1 = np.linalg.norm(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(y0 + h0 * f0, t0 + h0)
    d2 = np.linalg.norm((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / (d1 + d2)) ** (1.0 / (order + 1.0))
    return min(100.0 * h0, h1)


def mean_error_ratio(err, rtol, atol, y0, y1):
    tol = atol + rtol * np.maximum(np.abs(y0), np.abs(y1))
    return np.sqrt(np.mean((err / tol) ** 2))


def optimal_step_size(last_step, ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5.0):
    if ratio == 0:
        return last_step * ifactor
    dfac = 1.0 if ratio < 1 else dfactor
    factor = min(ifactor, max(ratio ** (-1.0 / order) * safety, dfac))
    return last_step * factor


def odeint(func, y0, ts, rtol=1.4e-8, atol=1.4e-8, mxstep=np.inf, hmax=np.inf, stats=None):
    """Adaptive Dopri5 with dense output at ``ts`` (ts[0] = initial time).  ``func(y, t) -> dy``
    on flat float64 arrays.  ``stats`` (dict) receives attempted/accepted step counts."""
    y0 = np.asarray(y0, dtype=np.float64).ravel()
    ts = np.asarray(ts, dtype=np.float64)
    f = func(y0, ts[0])
    dt = float(np.clip(initial_step_size(func, ts[0], y0, 4, rtol, atol, f), 0.0, hmax))
    y, t, last_t = y0, ts[0], ts[0]
    coeff = np.stack([y0] * 5)
    out = [y0]
    n_try = n_acc = 0
    acc_times = [float(ts[0])]
    for target in ts[1:]:
        i = 0
        while t < target and i < mxstep and dt > 0:
            y1, f1, err, k = runge_kutta_step(func, y, f, t, dt)
            ratio = mean_error_ratio(err, rtol, atol, y, y1)
            new_dt = float(np.clip(optimal_step_size(dt, ratio), 0.0, hmax))
            n_try += 1
            i += 1
            if ratio <= 1.0:
                coeff = interp_fit_dopri(y, y1, k, dt)
                y, f, last_t, t = y1, f1, t, t + dt
                n_acc += 1
                acc_times.append(float(t))
            dt = new_dt
        rel = (target - last_t) / (t - last_t)
        out.append(np.polyval(coeff, rel) if coeff.ndim == 1 else
                   ((((coeff[0] * rel + coeff[1]) * rel + coeff[2]) * rel + coeff[3]) * rel + coeff[4]))
    if stats is not None:
        stats.update(attempted=n_try, accepted=n_acc, step_times=np.array(acc_times))
    return np.stack(out)


def odeint_fixed(func, y0, ts, steps_per_interval, tableau="dopri5", stats=None, step_times=None):
    """Fixed-step explicit RK on the grid the MI355X engine uses: every output interval
    [ts[i], ts[i+1]] is split into ``steps_per_interval`` (one int, or one per interval) equal steps; outputs are step ends.
    ``dopri5``: 5th-order solution weights, 6 RHS evaluations per step with FSAL reuse.
    ``rk4``: classical 4-stage method."""
    y = np.asarray(y0, dtype=np.float64).ravel().copy()
    ts = np.asarray(ts, dtype=np.float64)
    out = [y.copy()]
    n = 0
    f = func(y, ts[0]) if tableau == "dopri5" else None
    spis = np.broadcast_to(np.asarray(steps_per_interval, dtype=np.int64), (max(len(ts) - 1, 0),))   # one count, or one per interval
    for a, b, spi in zip(ts[:-1], ts[1:], spis):
        for s in range(int(spi)):
            if step_times is None:
                h = (b - a) / int(spi)
                t = a + s * h
            else:       # caller-chosen step boundaries
                t, h = step_times[n], step_times[n + 1] - step_times[n]
            if tableau == "dopri5":
                y, f, _, _ = runge_kutta_step(func, y, f, t, h)
            else:
                k1 = func(y, t)
                k2 = func(y + 0.5 * h * k1, t + 0.5 * h)
                k3 = func(y + 0.5 * h * k2, t + 0.5 * h)
                k4 = func(y + h * k3, t + h)
                y = y + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
            n += 1
        out.append(y.copy())
    if stats is not None:
        stats.update(attempted=n, accepted=n)
    return np.stack(out)
