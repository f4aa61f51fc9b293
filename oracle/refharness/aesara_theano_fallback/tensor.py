"""
TEST INFRASTRUCTURE ONLY.  Eager-NumPy versions of the ``theano.tensor``
functions the reference calls (list obtained by grepping the reference for
``tt.<name>``; see oracle/refharness/README.md).  Our own code.
"""
import sys
import types

import numpy as np
import scipy.linalg

from . import T, Op, Apply, Placeholder, _Type, _wrap, _raw

_this = sys.modules[__name__]


def as_tensor_variable(x, name=None, ndim=None):
    if isinstance(x, T):
        return x
    if isinstance(x, (list, tuple)):
        x = np.array([np.asarray(v) for v in x])
    return T(x)


class TensorType(object):
    def __init__(self, dtype="float64", broadcastable=()):
        self.dtype = dtype
        self.broadcastable = tuple(broadcastable)
        self.ndim = len(self.broadcastable)

    def __call__(self, *args, **kwargs):
        return Placeholder(_Type(self.dtype, self.ndim))


def dvector(*a, **k):
    return Placeholder(_Type("float64", 1))


def dmatrix(*a, **k):
    return Placeholder(_Type("float64", 2))


def matrix(*a, dtype="float64", **k):
    return Placeholder(_Type(dtype, 2))


def _u1(fn):
    def f(x, *args, **kwargs):
        return _wrap(fn(np.asarray(x), *args, **kwargs))

    return f


def _u2(fn):
    def f(a, b):
        return _wrap(fn(np.asarray(a), np.asarray(b)))

    return f


exp = _u1(np.exp)
log = _u1(np.log)
cos = _u1(np.cos)
sin = _u1(np.sin)
sqrt = _u1(np.sqrt)
arctan = _u1(np.arctan)
floor = _u1(np.floor)
abs_ = _u1(np.abs)
isnan = _u1(np.isnan)
transpose = _u1(np.transpose)
mod = _u2(np.mod)
le = _u2(np.less_equal)
lt = _u2(np.less)
gt = _u2(np.greater)
ge = _u2(np.greater_equal)
eq = _u2(np.equal)
or_ = _u2(np.logical_or)
outer = _u2(np.outer)
maximum = _u2(np.maximum)
minimum = _u2(np.minimum)


def dot(a, b):
    return _wrap(np.dot(np.asarray(a), np.asarray(b)))


def tensordot(a, b, axes=2):
    return _wrap(np.tensordot(np.asarray(a), np.asarray(b), axes=axes))


def batched_dot(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.ndim == 2 and b.ndim == 2:
        return _wrap(np.einsum("bk,bk->b", a, b))
    return _wrap(np.einsum("b...k,bk...->b...", a, b))


def reshape(x, shape, ndim=None):
    shape = tuple(int(s) for s in np.atleast_1d(_raw(shape)))
    return _wrap(np.reshape(np.asarray(x), shape))


def sum(x, axis=None, **kwargs):  # noqa: A001
    if isinstance(x, (list, tuple)):
        x = np.array([np.asarray(v) for v in x])
    return _wrap(np.sum(np.asarray(x), axis=axis))


def mean(x, axis=None, **kwargs):
    return _wrap(np.mean(np.asarray(x), axis=axis))


def argmax(x, axis=None, **kwargs):
    return _wrap(np.argmax(np.asarray(x), axis=axis))


def switch(c, a, b):
    return _wrap(np.where(np.asarray(c), np.asarray(a), np.asarray(b)))


def cast(x, dtype):
    return _wrap(np.asarray(x).astype(dtype))


def shape(x):
    return np.asarray(x).shape


def _shp(s):
    if isinstance(s, (int, np.integer)):
        return (int(s),)
    return tuple(int(v) for v in s)


def zeros(shape, dtype="float64"):
    return T(np.zeros(_shp(shape), dtype=dtype))


def ones(shape, dtype="float64"):
    return T(np.ones(_shp(shape), dtype=dtype))


def zeros_like(x, **k):
    return T(np.zeros_like(np.asarray(x)))


def ones_like(x, **k):
    return T(np.ones_like(np.asarray(x)))


def eye(n, m=None, k=0, dtype="float64"):
    return T(np.eye(int(n), None if m is None else int(m), k, dtype=dtype))


def arange(*args, **kwargs):
    return T(np.arange(*[np.asarray(a)[()] for a in args], **kwargs))


def diag(x, k=0):
    return _wrap(np.diag(np.asarray(x), k))


def tril(x, k=0):
    return _wrap(np.tril(np.asarray(x), k))


def triu(x, k=0):
    return _wrap(np.triu(np.asarray(x), k))


def tile(x, reps, ndim=None):
    return _wrap(np.tile(np.asarray(x), _shp(reps)))


def swapaxes(x, a, b):
    return _wrap(np.swapaxes(np.asarray(x), a, b))


def concatenate(xs, axis=0):
    return _wrap(np.concatenate([np.atleast_1d(np.asarray(v)) for v in xs], axis))


def _sub(x, y, inc):
    parent = getattr(x, "_parent", None)
    if parent is None:
        raise ValueError("set_subtensor needs the result of an indexing op")
    base, idx = parent
    out = np.array(np.asarray(base), copy=True)
    if inc:
        out[idx] += np.asarray(y)
    else:
        out[idx] = np.asarray(y)
    return T(out)


def set_subtensor(x, y, **kwargs):
    return _sub(x, y, False)


def inc_subtensor(x, y, **kwargs):
    return _sub(x, y, True)


class ExtractDiag(object):
    def __init__(self, offset=0, axis1=0, axis2=1, view=False):
        self.offset, self.axis1, self.axis2 = offset, axis1, axis2

    def __call__(self, x):
        return _wrap(
            np.diagonal(np.asarray(x), self.offset, self.axis1, self.axis2)
        )


# extra_ops ----------------------------------------------------------------
class _CpuContiguous(object):
    def __call__(self, x):
        return _wrap(np.ascontiguousarray(np.asarray(x)))


extra_ops = types.SimpleNamespace(CpuContiguous=_CpuContiguous)


# nlinalg ------------------------------------------------------------------
class _Eig(Op):
    pass


nlinalg = types.SimpleNamespace(Eig=_Eig)


# slinalg (bases only: the reference overrides `perform`, math.py:20-91) -----
class _Cholesky(Op):
    __props__ = ("lower", "destructive", "on_error")

    def __init__(self, lower=True, on_error="raise"):
        self.lower = lower
        self.destructive = False
        self.on_error = on_error

    def make_node(self, x):
        x = as_tensor_variable(x)
        assert x.ndim == 2
        return Apply(self, [x], [x.type()])

    def perform(self, node, inputs, outputs):
        outputs[0][0] = scipy.linalg.cholesky(inputs[0], lower=self.lower)


class _Solve(Op):
    __props__ = ("A_structure", "lower", "overwrite_A", "overwrite_b")

    def __init__(
        self,
        A_structure="general",
        lower=False,
        overwrite_A=False,
        overwrite_b=False,
    ):
        self.A_structure = A_structure
        self.lower = lower
        self.overwrite_A = overwrite_A
        self.overwrite_b = overwrite_b

    def make_node(self, A, b):
        A = as_tensor_variable(A)
        b = as_tensor_variable(b)
        assert A.ndim == 2
        assert b.ndim in (1, 2)
        return Apply(self, [A, b], [b.type()])


slinalg = types.ModuleType(__name__ + ".slinalg")
slinalg.Cholesky = _Cholesky
slinalg.Solve = _Solve
sys.modules[slinalg.__name__] = slinalg
