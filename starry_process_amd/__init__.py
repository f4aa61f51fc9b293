"""MI355X-native log-likelihood hot path of starry_process."""
__version__ = "0.1.0"

import os as _os

# One hardware queue per HIP stream in flight: the runtime reads this when it initialises and maps streams onto 4 queues
# by default -- the four streams of EnsembleLogProb / EnsembleGradient / engine_slots then share queues and lose ~10 %.
# Harmless for everyone else; a value the caller has set wins.  (engine.engine_slots still warns when the runtime is
# already up with fewer queues than streams asked for.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
