"""
Counterpart of the reference's ``math.py``: the NaN-tolerant ``Cholesky`` / ``Solve`` Ops
with their reverse mode, ``cho_factor`` / ``cho_solve`` (math.py:20-100), ``cast`` and
``matrix_sqrt`` (math.py:103-139), eager and on the GPU.

    L = cho_factor(C)                    # all-NaN when C is not positive definite
    x = cho_solve(L, b)
    [C_bar] = cho_factor.L_op([C], [L], [L_bar])
    [A_bar, b_bar] = Solve("lower_triangular", lower=True).L_op([L, b], [c], [c_bar])

``L_op`` keeps Theano's calling convention (lists of inputs, outputs, output gradients)
so code written against the reference's Ops reads the same.  NumPy in -> NumPy out, torch
CUDA tensors in -> torch CUDA tensors out; every number comes from ``libsp_hip.so``
(sp_cho_factor, sp_tri_solve, sp_solve_rev, sp_cholesky_rev) -- there is no CPU path.
"""
import numpy as np

from .defaults import defaults
from .engine import get_engine
from .ops import _is_torch, _out
from .upstream import matrix_sqrt

__all__ = ["is_tensor", "cho_solve", "cho_factor", "cast", "matrix_sqrt", "Solve", "Cholesky"]


def _engine():
    return get_engine(defaults["ydeg"], defaults["udeg"])


def is_tensor(*objs):
    """True if any of ``objs`` lives on the device (the eager stand-in for "is a Theano
    variable", math.py:11-17)."""
    return any(_is_torch(o) for o in objs)


class Solve(object):
    """Triangular solve c = A^-1 b with A = L (``lower_triangular``) or A = L^T
    (``upper_triangular``); NaN in -> NaN out (math.py:20-38)."""

    def __init__(self, A_structure="lower_triangular", lower=None):
        if A_structure not in ("lower_triangular", "upper_triangular"):
            raise ValueError("only triangular systems are on this path (math.py:97-100)")
        self.A_structure = A_structure
        self.lower = A_structure == "lower_triangular" if lower is None else bool(lower)

    def _L(self, A):
        """The lower factor the device routines read: A itself, or A^T for an upper system."""
        if self.A_structure == "lower_triangular":
            return A
        return A.transpose(-1, -2) if _is_torch(A) else np.swapaxes(np.asarray(A), -1, -2)

    def __call__(self, A, b):
        e = _engine()
        tt = _is_torch(b)
        trans = self.A_structure == "upper_triangular"
        return _out(e.tri_solve(e.f64(self._L(A)).contiguous(), b, trans=trans), tt)

    def L_op(self, inputs, outputs, output_gradients):
        """[A_bar, b_bar] (math.py:40-72)."""
        A, b = inputs
        c = outputs[0]
        c_bar = output_gradients[0]
        e = _engine()
        tt = _is_torch(c_bar)
        trans = self.A_structure == "upper_triangular"
        A_bar, b_bar = e.solve_rev(e.f64(self._L(A)).contiguous(), c, c_bar, trans=trans)
        return [_out(A_bar, tt), _out(b_bar, tt)]


class Cholesky(object):
    """Lower Cholesky factor; a matrix that is not positive definite gives an all-NaN factor
    when ``on_error="nan"`` (math.py:75-91) and raises otherwise."""

    def __init__(self, lower=True, on_error="raise"):
        if not lower:
            raise ValueError("the path factors lower triangles only (math.py:94)")
        self.lower = True
        self.on_error = on_error

    def __call__(self, x):
        e = _engine()
        tt = _is_torch(x)
        L, info = e.cho_factor(x)
        if self.on_error != "nan" and int(info.max().item()) != 0:
            raise np.linalg.LinAlgError("matrix is not positive definite")
        return _out(L, tt)

    def L_op(self, inputs, outputs, gradients):
        """[C_bar] from the factor and its gradient."""
        e = _engine()
        tt = _is_torch(gradients[0])
        return [_out(e.cholesky_rev(outputs[0], gradients[0]), tt)]


cho_factor = Cholesky(on_error="nan")


def cho_solve(cho_A, b):
    solve_lower = Solve(A_structure="lower_triangular", lower=True)
    solve_upper = Solve(A_structure="upper_triangular", lower=False)
    At = cho_A.transpose(-1, -2) if _is_torch(cho_A) else np.swapaxes(np.asarray(cho_A), -1, -2)
    return solve_upper(At, solve_lower(cho_A, b))


def cast(*args, vectorize=False):
    """float64 arrays (flattened when ``vectorize``), math.py:103-118."""
    out = [np.asarray(a, dtype="float64") if not _is_torch(a) else a.double() for a in args]
    if vectorize:
        out = [a.reshape(-1) for a in out]
    return out[0] if len(out) == 1 else out
