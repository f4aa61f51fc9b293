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


def ExpSquaredKernel(t1, t2, tau):
    dt = np.abs(np.reshape(t1, (-1, 1)) - np.reshape(t2, (1, -1)))
    return np.exp(-(dt ** 2) / (2 * tau))


def Matern32Kernel(t1, t2, tau):
    dt = np.abs(np.reshape(t1, (-1, 1)) - np.reshape(t2, (1, -1)))
    x = np.sqrt(3) * dt / tau
    return (1 + x) * np.exp(-x)


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
