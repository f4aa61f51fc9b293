"""
Eager counterparts of the reference's Theano Ops for the log-likelihood path
(SURVEY.md 8b).  Same names and call arity as ``starry_process.ops``:

    RxOp(ydeg)(theta)                        -> (R, dR/dtheta)      ops/wigner/Rx.py:8-43
    tensordotRzOp(ydeg)(M, theta)            -> f                   ops/wigner/tensordotRz.py:9-37
    special_tensordotRzOp(ydeg)(T, M, theta) -> f                   ops/wigner/special_tensordotRz.py:9-37
    rTA1Op(ydeg)()                           -> rTA1                ops/flux/rTA1.py:8-21
    rTA1LOp(ydeg, udeg)(u)                   -> rTA1L               ops/flux/rTA1L.py:8-43
    AlphaBetaOp(N)(z)                        -> (alpha, beta, dalpha/dz, dbeta/dz)   ops/norm/norm.py:8-44
    CheckBoundsOp(name, lower, upper)(x)     -> x or ValueError     ops/exceptions.py:8-56
    cho_factor(A), cho_solve(L, b)                                   math.py:75-100

Inputs may be NumPy arrays or torch tensors; NumPy in -> NumPy out (results are
copied back to the host), torch CUDA tensors in -> torch CUDA tensors out.
Every Op runs on the GPU through the C ABI; there is no CPU implementation.
Results carry an ``eval()`` method so code written for the lazy reference
(``op(x).eval()``) runs unchanged.
"""
import numpy as np

from . import _lib
from .defaults import defaults
from .engine import get_engine

__all__ = [
    "RxOp", "tensordotRzOp", "special_tensordotRzOp", "rTA1Op", "rTA1LOp",
    "AlphaBetaOp", "CheckBoundsOp", "CheckVectorSizeOp", "cho_factor", "cho_solve",
    "Eager",
]


class Eager(np.ndarray):
    """ndarray with a no-op ``eval()`` (stands in for a Theano variable)."""

    def __new__(cls, value):
        return np.asarray(value).view(cls)

    def eval(self, *args, **kwargs):
        out = np.asarray(self)
        return out[()] if out.ndim == 0 else out


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _out(t, like_torch):
    return t if like_torch else Eager(t.detach().cpu().numpy())


class _BaseOp(object):
    def __init__(self, ydeg=defaults["ydeg"], udeg=defaults["udeg"], compile_args=None, **kwargs):
        self.ydeg = int(ydeg)
        self.udeg = int(udeg)
        self.N = (self.ydeg + 1) ** 2

    @property
    def engine(self):
        return get_engine(self.ydeg, self.udeg)


class RxOp(_BaseOp):
    def __call__(self, theta):
        if np.ndim(theta) != 0:
            raise ValueError("theta must be a scalar")
        R, dR = self.engine.Rx([float(theta)])
        return Eager(R[0].cpu().numpy()), Eager(dR[0].cpu().numpy())


class tensordotRzOp(_BaseOp):
    def __call__(self, M, theta):
        tt = _is_torch(M)
        if np.ndim(M) != 2:
            raise ValueError("M must be a matrix")
        if np.ndim(theta) != 1:
            raise ValueError("theta must be a vector")
        return _out(self.engine.tensordotRz(M, theta), tt)

    def grad(self, inputs, gradients):
        """[bM, btheta] (ops/wigner/tensordotRz.py:33-34 -> tensordotRzRevOp)."""
        M, theta = inputs
        tt = _is_torch(M)
        bM, bth = self.engine.tensordotRz_rev(M, theta, gradients[0])
        return [_out(bM, tt), _out(bth, tt)]


class special_tensordotRzOp(_BaseOp):
    def __call__(self, T, M, theta):
        tt = _is_torch(M)
        if np.ndim(T) != 2:
            raise ValueError("T must be a matrix")
        if np.ndim(M) != 2:
            raise ValueError("M must be a matrix")
        if np.ndim(theta) != 1:
            raise ValueError("theta must be a vector")
        return _out(self.engine.special_tensordotRz(T, M, theta), tt)

    def grad(self, inputs, gradients):
        """[zeros(N, N), bM, btheta]: like the reference, no gradient flows to T
        (ops/wigner/special_tensordotRz.py:29-32 -> special_tensordotRzRevOp)."""
        T, M, theta = inputs
        tt = _is_torch(M)
        bM, bth = self.engine.special_tensordotRz_rev(T, M, theta, gradients[0])
        zero = bM * 0.0
        return [_out(zero, tt), _out(bM, tt), _out(bth, tt)]


class rTA1Op(_BaseOp):
    def __call__(self):
        return Eager(self.engine.rTA1())


class rTA1LOp(_BaseOp):
    def __call__(self, u):
        if np.ndim(u) != 1:
            raise ValueError("u must be a vector")
        return Eager(self.engine.rTA1L(np.asarray(u, dtype=float)[: self.udeg])[0])

    def grad(self, inputs, gradients):
        """(bu,) (ops/flux/rTA1L.py:27-28 -> rTA1LRevOp)."""
        return (Eager(self.engine.rTA1L_rev(np.asarray(inputs[0], dtype=float), np.asarray(gradients[0], dtype=float))),)


class AlphaBetaOp(object):
    def __init__(self, N=20):
        self.N = N

    def __call__(self, z):
        return tuple(Eager(v) for v in _lib.alpha_beta(float(z), self.N))


class CheckBoundsOp(object):
    def __init__(self, lower=-np.inf, upper=np.inf, name=None, tol=1e-6):
        self.lower, self.upper, self.tol = lower, upper, tol
        self.name = "parameter" if name is None else name

    def __call__(self, x):
        v = np.asarray(x, dtype=float)
        low = v < self.lower - self.tol
        high = v > self.upper + self.tol
        if np.any(low | high):
            if np.any(low):
                value, sign, bound = np.atleast_1d(v)[np.where(np.atleast_1d(low))[0][0]], "<=", self.lower
            else:
                value, sign, bound = np.atleast_1d(v)[np.where(np.atleast_1d(high))[0][0]], ">=", self.upper
            raise ValueError("%s out of bounds: %f %s %f" % (self.name, value, sign, bound))
        return x


class CheckVectorSizeOp(object):
    def __init__(self, name=None, size=None):
        self.size = size
        self.name = "vector" if name is None else name

    def __call__(self, x):
        if np.size(x) != self.size:
            raise ValueError(
                "Vector `%s` has the wrong size. Expected %d, got %d." % (self.name, self.size, np.size(x))
            )
        return x


def cho_factor(A, ydeg=None):
    """Lower Cholesky factor; all-NaN when A is not positive definite
    (math.py:75-94)."""
    tt = _is_torch(A)
    e = get_engine(defaults["ydeg"] if ydeg is None else ydeg, defaults["udeg"])
    L, _ = e.cho_factor(A)
    return _out(L, tt)


def cho_solve(cho_A, b, ydeg=None):
    """(L L^T)^-1 b (math.py:97-100)."""
    tt = _is_torch(b)
    e = get_engine(defaults["ydeg"] if ydeg is None else ydeg, defaults["udeg"])
    return _out(e.cho_solve(cho_A, b), tt)
