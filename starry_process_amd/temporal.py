"""
Temporal covariance kernels (reference ``temporal.py:8-16``).

On the device the two kernels are evaluated inside the covariance-assembly
kernels (``csrc/sp_assemble.hip``, ``temporal_factor``); the callables below are
the names user scripts pass as ``temporal_kernel=`` and they also work as plain
host functions on NumPy arrays.  ``kernel_id`` maps a callable to the device
kernel selector.
"""
import numpy as np

__all__ = ["ExpSquaredKernel", "Matern32Kernel", "kernel_id"]


def _lags(t1, t2):
    """|t1_i - t2_j| for every pair, as a (len(t1), len(t2)) array."""
    return np.abs(np.subtract.outer(np.ravel(t1), np.ravel(t2)))


def ExpSquaredKernel(t1, t2, tau):
    """exp(-dt^2 / (2 tau)): ``tau`` enters linearly, as in the reference (not tau^2)."""
    lag = _lags(t1, t2)
    return np.exp(-0.5 * lag * lag / tau)


def Matern32Kernel(t1, t2, tau):
    """(1 + s) exp(-s), s = sqrt(3) dt / tau."""
    s = _lags(t1, t2) * (np.sqrt(3) / tau)
    return (1 + s) * np.exp(-s)


def kernel_id(kernel):
    if kernel is None:
        return None
    if kernel is Matern32Kernel or getattr(kernel, "__name__", "") == "Matern32Kernel":
        return "matern32"
    if kernel is ExpSquaredKernel or getattr(kernel, "__name__", "") == "ExpSquaredKernel":
        return "expsquared"
    raise NotImplementedError(
        "only Matern32Kernel and ExpSquaredKernel are implemented on the device"
    )
