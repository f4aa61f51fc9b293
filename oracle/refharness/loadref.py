"""
TEST INFRASTRUCTURE ONLY (development container only).

Loads the reference package *from where it lies* (``/root/reference``) on top
of the eager stand-in in ``aesara_theano_fallback/``, skipping the reference's
``__init__.py`` (which imports pymc3, absent here).  Nothing is copied.

    from oracle.refharness.loadref import load_reference
    ref = load_reference()          # ref.sp.StarryProcess, ref.flux, ...
"""
import importlib
import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("SP_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "starry_process"))


def load_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import aesara_theano_fallback  # noqa: F401  (the eager stand-in)

    if "starry_process" not in sys.modules or not hasattr(
        sys.modules["starry_process"], "_sp_refharness"
    ):
        pkg = types.ModuleType("starry_process")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "starry_process")]
        pkg._sp_refharness = True
        pkg.CACHE_DEV_C_CODE = False
        sys.modules["starry_process"] = pkg
        ver = types.ModuleType("starry_process.starry_process_version")
        ver.__version__ = "0.0.0.dev0"
        sys.modules[ver.__name__] = ver
    ns = types.SimpleNamespace()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in (
            "compat",
            "defaults",
            "temporal",
            "wigner",
            "math",
            "ops",
            "integrals",
            "size",
            "latitude",
            "longitude",
            "contrast",
            "flux",
            "sp",
        ):
            setattr(ns, name, importlib.import_module("starry_process." + name))
    return ns
