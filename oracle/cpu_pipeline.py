"""
TEST / BENCH INFRASTRUCTURE -- Python side of oracle/cpu_pipeline.c: the CPU restatement of the
per-star marginal likelihood pipeline in C with OpenMP over stars.

The per-hyperparameter-sample part (polar moments, inclination integrals, kernel table) is the
NumPy oracle's (oracle/sp_oracle.py); LAPACK's dpotrf / dtrtrs come from the SciPy that runs the
oracle (scipy.linalg.cython_lapack's C function pointers).  Never touches the GPU library.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "libcpupipe.so"])


def _lapack_ptr(name):
    from scipy.linalg import cython_lapack

    cap = cython_lapack.__pyx_capi__[name]
    ctypes.pythonapi.PyCapsule_GetName.restype = ctypes.c_char_p
    ctypes.pythonapi.PyCapsule_GetName.argtypes = [ctypes.py_object]
    ctypes.pythonapi.PyCapsule_GetPointer.restype = ctypes.c_void_p
    ctypes.pythonapi.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
    return ctypes.c_void_p(ctypes.pythonapi.PyCapsule_GetPointer(cap, ctypes.pythonapi.PyCapsule_GetName(cap)))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcpupipe.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.sp_cpu_lnlike.restype = ctypes.c_int
    return _LIB


def table_for(op, u=(0.0, 0.0)):
    """(tab [5, covpts + 4], mean, var) of an oracle.OracleProcess for limb darkening u."""
    from . import sp_oracle as orc

    rta1 = op._rta1(u)
    w, W = orc.inclination_integrals(op.ydeg, rta1)
    mean, var = orc.marginal_mean_var(op.ydeg, w, W, op.ez, op.Ez)
    tb = orc.kernel_table(op.ydeg, W, op.Ez, mean, op.covpts)
    np_ = op.covpts + 4
    tab = np.zeros((5, np_))
    tab[0] = tb["xp"]
    for k, name in enumerate(("a0", "a1", "a2", "a3")):
        tab[k + 1, : tb[name].shape[0]] = tb[name]
    return tab, float(mean), float(var)


def lnlike(op, t, flux, period, data_var, u=(0.0, 0.0), nthreads=0):
    """Log-likelihoods of S stars (t, flux: [S, K]) under oracle process `op` (marginal case).
    Returns (values [S], seconds, threads)."""
    t = np.ascontiguousarray(np.asarray(t, dtype=np.float64))
    flux = np.ascontiguousarray(np.asarray(flux, dtype=np.float64))
    S, K = flux.shape
    period = np.ascontiguousarray(np.broadcast_to(np.asarray(period, dtype=np.float64), (S,)))
    data_var = np.ascontiguousarray(np.broadcast_to(np.asarray(data_var, dtype=np.float64), (S,)))
    tab, mean, var = table_for(op, u)
    tab = np.ascontiguousarray(tab)
    out = np.empty(S)
    secs = ctypes.c_double()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    used = lib().sp_cpu_lnlike(
        ctypes.c_int(S), ctypes.c_int(K), P(t), P(flux), P(period), P(data_var), ctypes.c_int(op.covpts),
        P(tab), ctypes.c_double(mean), ctypes.c_double(var), ctypes.c_int(1 if op.normalized else 0),
        ctypes.c_int(op.normN), ctypes.c_double(op.zmax), _lapack_ptr("dpotrf"), _lapack_ptr("dtrtrs"),
        ctypes.c_int(int(nthreads)), P(out), ctypes.byref(secs))
    return out, secs.value, used
