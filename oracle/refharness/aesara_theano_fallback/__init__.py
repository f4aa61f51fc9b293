"""
TEST INFRASTRUCTURE ONLY (development container only; never shipped to the
GPU box as part of the product and never imported by ``starry_process_amd``).

An *eager NumPy* stand-in for the small slice of the Theano/Aesara API that the
reference ``starry_process`` package touches (reference ``compat.py:2-6``).
With it on ``sys.path`` the reference's UNMODIFIED Python sources (``sp.py``,
``flux.py``, ``math.py``, ``integrals.py`` ...) execute directly from
``/root/reference`` and every "symbolic" expression is evaluated immediately,
so the reference itself produces the golden vectors under ``tests/golden``.

Native ops (``ExternalCOp`` subclasses with a ``func_name``) are dispatched to
``oracle/_ref/libspref_L*_U*.so`` which is the reference's own C++ compiled
from its own headers (``oracle/refharness/refshim.cc``).

This file is our own code; it contains no reference source.
"""
import ctypes
import os
import sys
import types

import numpy as np

USE_AESARA = True

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFDIR = os.path.abspath(os.path.join(_HERE, "..", "..", "_ref"))


# --------------------------------------------------------------------------
# Eager tensor
# --------------------------------------------------------------------------
class _Type(object):
    """What ``x.type`` returns: calling it makes an output placeholder."""

    def __init__(self, dtype="float64", ndim=None):
        self.dtype = dtype
        self.ndim = ndim

    def __call__(self, *args, **kwargs):
        return Placeholder(self)


class Placeholder(object):
    """Output slot created in ``make_node``; filled by ``perform``."""

    def __init__(self, type_=None):
        self.type = type_ if type_ is not None else _Type()
        self.dtype = self.type.dtype


class Node(object):
    """Common base of everything "symbolic" (reference math.py:11-18)."""


class T(np.ndarray, Node):
    """ndarray that quacks like a Theano variable, evaluated eagerly."""

    _parent = None

    def __new__(cls, value, dtype=None):
        arr = np.array(value, dtype=dtype, copy=True)
        return arr.view(cls)

    def __array_finalize__(self, obj):
        self._parent = None

    # Theano variables compare by identity (flux.py:243-248 relies on it)
    def __eq__(self, other):
        return self is other

    def __ne__(self, other):
        return self is not other

    __hash__ = object.__hash__

    # In Theano `x += y` rebinds, it never mutates the operand
    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o

    def __getitem__(self, idx):
        out = np.ndarray.__getitem__(self, idx)
        if not isinstance(out, T):
            out = T(out)
        else:
            out = T(np.asarray(out))
        out._parent = (self, idx)
        return out

    def eval(self, *args, **kwargs):
        return np.array(self, copy=True)

    def astype(self, dtype, **kwargs):
        return T(np.asarray(self).astype(dtype))

    @property
    def type(self):
        return _Type(str(self.dtype), self.ndim)

    def dot(self, other):
        return T(np.dot(np.asarray(self), np.asarray(other)))

    def zeros_like(self):
        return T(np.zeros_like(np.asarray(self)))

def _wrap(x):
    if isinstance(x, T):
        return x
    return T(x)


def _raw(x):
    if isinstance(x, (list, tuple)):
        return [_raw(v) for v in x]
    return np.asarray(x)


# --------------------------------------------------------------------------
# graph: Apply / Op / ExternalCOp / Params
# --------------------------------------------------------------------------
class Apply(object):
    def __init__(self, op, inputs, outputs):
        self.op = op
        self.inputs = list(inputs)
        self.outputs = list(outputs)


_LIBS = {}


def _reflib(ydeg, udeg):
    key = (int(ydeg), int(udeg))
    if key not in _LIBS:
        path = os.path.join(_REFDIR, "libspref_L%d_U%d.so" % key)
        if not os.path.exists(path):
            raise RuntimeError(
                "reference library %s missing: run `make -C oracle ref "
                "CONFIGS=%d_%d`" % (path, key[0], key[1])
            )
        lib = ctypes.CDLL(path)
        assert lib.spref_lmax() == key[0] and lib.spref_umax() == key[1]
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _dispatch_native(op, inputs):
    """Run one of the reference's C++ ops through oracle/_ref."""
    name = op.func_name
    lib = _reflib(op.ydeg, op.udeg)
    N = (op.ydeg + 1) ** 2
    nwig = ((op.ydeg + 1) * (2 * op.ydeg + 1) * (2 * op.ydeg + 3)) // 3
    if name == "APPLY_SPECIFIC(Rx)":
        theta = float(np.asarray(inputs[0]))
        R = np.empty(nwig)
        dR = np.empty(nwig)
        lib.spref_Rx(ctypes.c_double(theta), _p(R), _p(dR))
        return [R, dR]
    if name == "APPLY_SPECIFIC(tensordotRz)":
        M = _c(inputs[0])
        theta = _c(inputs[1])
        if M.ndim != 2 or theta.ndim != 1:
            raise ValueError("bad shapes")
        K = theta.shape[0]
        f = np.empty((K, N))
        lib.spref_tensordotRz(_p(M), _p(theta), ctypes.c_int(K), _p(f))
        return [f]
    if name == "APPLY_SPECIFIC(special_tensordotRz)":
        Tm = _c(inputs[0])
        M = _c(inputs[1])
        theta = _c(inputs[2])
        K = theta.shape[0]
        f = np.empty(K)
        lib.spref_special_tensordotRz(
            _p(Tm), _p(M), _p(theta), ctypes.c_int(K), _p(f)
        )
        return [f]
    if name == "APPLY_SPECIFIC(rTA1)":
        f = np.empty(N)
        lib.spref_rTA1(_p(f))
        return [f]
    if name == "APPLY_SPECIFIC(rTA1L)":
        u = _c(inputs[0])
        assert u.shape == (op.udeg,)
        f = np.empty(N)
        lib.spref_rTA1L(_p(u), _p(f))
        return [f]
    if name == "APPLY_SPECIFIC(tensordotRz_rev)":
        M, theta, bf = _c(inputs[0]), _c(inputs[1]), _c(inputs[2])
        K = theta.shape[0]
        bM, bth = np.empty((K, N)), np.empty(K)
        lib.spref_tensordotRz_rev(_p(M), _p(theta), ctypes.c_int(K), _p(bf), _p(bM), _p(bth))
        return [bM, bth]
    if name == "APPLY_SPECIFIC(special_tensordotRz_rev)":
        Tm, M, theta, bf = _c(inputs[0]), _c(inputs[1]), _c(inputs[2]), _c(inputs[3])
        K = theta.shape[0]
        bM, bth = np.empty((N, N)), np.empty(K)
        lib.spref_special_tensordotRz_rev(_p(Tm), _p(M), _p(theta), ctypes.c_int(K), _p(bf),
                                          _p(bM), _p(bth))
        return [bM, bth]
    if name == "APPLY_SPECIFIC(rTA1L_rev)":
        u, bf = _c(inputs[0]), _c(inputs[1])
        bu = np.empty(op.udeg)
        lib.spref_rTA1L_rev(_p(u), _p(bf), _p(bu))
        return [bu]
    if name == "APPLY_SPECIFIC(latitude)":
        alpha = float(np.asarray(inputs[0]))
        beta = float(np.asarray(inputs[1]))
        q, dqda, dqdb = np.empty(N), np.empty(N), np.empty(N)
        Q, dQda, dQdb = np.empty((N, N)), np.empty((N, N)), np.empty((N, N))
        lib.spref_latitude(
            ctypes.c_double(alpha),
            ctypes.c_double(beta),
            _p(q),
            _p(dqda),
            _p(dqdb),
            _p(Q),
            _p(dQda),
            _p(dQdb),
        )
        return [q, dqda, dqdb, Q, dQda, dQdb]
    raise NotImplementedError("native op %s not wired in the harness" % name)


class Op(object):
    __props__ = ()

    def make_node(self, *inputs):
        raise NotImplementedError

    def perform(self, node, inputs, output_storage):
        raise NotImplementedError

    def __call__(self, *inputs, **kwargs):
        node = self.make_node(*inputs)
        ins = [np.array(np.asarray(i), copy=True) for i in node.inputs]
        if getattr(self, "func_name", None) is not None:
            outs = _dispatch_native(self, ins)
        else:
            storage = [[None] for _ in node.outputs]
            self.perform(node, ins, storage)
            outs = [s[0] for s in storage]
        outs = [T(o) for o in outs]
        if len(outs) == 1:
            return outs[0]
        return outs


class ExternalCOp(Op):
    func_file = None
    func_name = None

    def __init__(self, func_files=None, func_name=None):
        pass


class Params(object):
    pass


class ParamsType(object):
    pass


graph = types.ModuleType("aesara_theano_fallback.graph")
graph.basic = types.SimpleNamespace(Node=Node, Apply=Apply)
graph.op = types.SimpleNamespace(Op=Op, ExternalCOp=ExternalCOp)
graph.params_type = types.SimpleNamespace(Params=Params, ParamsType=ParamsType)
graph.fg = types.SimpleNamespace()
sys.modules[graph.__name__] = graph


# --------------------------------------------------------------------------
# ifelse
# --------------------------------------------------------------------------
def ifelse(cond, a, b):
    pick = a if bool(np.all(np.asarray(cond))) else b
    if isinstance(pick, (int, float, np.ndarray)):
        return _wrap(pick)
    return pick


# --------------------------------------------------------------------------
# the `aesara` (a.k.a. `theano`) namespace
# --------------------------------------------------------------------------
class DisconnectedType(object):
    pass


class _Config(object):
    floatX = "float64"
    cast_policy = "numpy+floatX"
    compute_test_value = "off"


aesara = types.ModuleType("aesara")
aesara.config = _Config()
aesara.gradient = types.SimpleNamespace(DisconnectedType=DisconnectedType)
aesara.scalar = types.SimpleNamespace(
    upcast=lambda *dts: str(np.result_type(*dts))
)


def _function(inputs, outputs, **kwargs):
    raise NotImplementedError(
        "eager stand-in: expressions are already evaluated; call .eval()"
    )


aesara.function = _function


class RandomStream(object):
    """Eager RandomStream (reference sp.py:284, compat.py:40-48)."""

    def __init__(self, seed=0):
        self._rng = np.random.RandomState(seed)

    def normal(self, size=None, **kwargs):
        return T(self._rng.normal(size=tuple(int(s) for s in size)))

    def uniform(self, size=None, **kwargs):
        return T(self._rng.uniform(size=tuple(int(s) for s in size)))


for _name in (
    "aesara.tensor",
    "aesara.tensor.random",
    "aesara.tensor.random.utils",
):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["aesara"] = aesara
sys.modules["aesara.tensor.random.utils"].RandomStream = RandomStream
aesara.tensor = sys.modules["aesara.tensor"]

from . import tensor  # noqa: E402  (needs T / Op defined above)
