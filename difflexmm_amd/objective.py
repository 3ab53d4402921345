"""Cross-correlation measures between recorded signals -- the names of ``difflexmm/objective.py`` (post-processing of space-time
records in the notebooks; host code, SciPy where the reference uses ``jax.scipy.signal``)."""
import numpy as np
import scipy.signal


def compute_xcorr2d(signal0, signal1, shift=(None, None)):
    """objective.py:10-39: full 2-D cross-correlation of two 2-D arrays divided by the peak of signal0's auto-correlation; with a shift
    along one axis (0 = no shift) the corresponding slice, with both the single value."""
    signal0, signal1 = np.asarray(signal0, dtype=float), np.asarray(signal1, dtype=float)
    xcorr2d = scipy.signal.correlate2d(signal0, signal1) / scipy.signal.correlate2d(signal0, signal0).max()
    s0, s1 = shift
    if s0 is None and s1 is None:
        return xcorr2d
    if s1 is None:
        return xcorr2d[signal1.shape[0] - 1 + s0, :]
    if s0 is None:
        return xcorr2d[:, signal1.shape[1] - 1 + s1]
    return xcorr2d[signal1.shape[0] - 1 + s0, signal1.shape[1] - 1 + s1]


def compute_xcorr(signal0, signal1, shift=None):
    """objective.py:42-57: the 1-D counterpart."""
    signal0, signal1 = np.asarray(signal0, dtype=float), np.asarray(signal1, dtype=float)
    xcorr = scipy.signal.correlate(signal0, signal1) / scipy.signal.correlate(signal0, signal0).max()
    return xcorr if shift is None else xcorr[signal1.shape[0] - 1 + shift]


def compute_max_xcorr2d_at_shift(signal0, signal1, shift, shift_axis=0):
    """objective.py:60-75: (maximum of the slice at ``shift`` along ``shift_axis``, delay along the other axis; delay > 0: signal1 lags)."""
    signal1 = np.asarray(signal1)
    sl = compute_xcorr2d(signal0, signal1, shift=(shift, None) if shift_axis == 0 else (None, shift))
    return sl.max(), -(int(sl.argmax()) + 1 - signal1.shape[1 if shift_axis == 0 else 0])


def compute_space_time_xcorr(space_time0, space_time1):
    """objective.py:78-89: space on axis 0, time on axis 1 -> (largest cross-correlation at zero space shift, its time delay)."""
    return compute_max_xcorr2d_at_shift(space_time0, space_time1, shift=0, shift_axis=0)
