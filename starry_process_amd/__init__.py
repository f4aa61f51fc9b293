"""MI355X-native log-likelihood hot path of starry_process."""
__version__ = "0.1.0"

from .defaults import defaults  # noqa: F401
from .temporal import ExpSquaredKernel, Matern32Kernel  # noqa: F401


def __getattr__(name):
    # heavy modules (torch, the HIP library) load on first use
    if name in ("StarryProcess", "StarryProcessSum"):
        from . import sp

        return getattr(sp, name)
    if name in ("gauss2beta", "beta2gauss"):
        from . import upstream

        return getattr(upstream, name)
    if name in ("ops", "flux", "sp", "engine", "ensemble", "upstream", "upstream_device", "hostconst",
                "calibrate", "math", "grad"):
        import importlib

        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
