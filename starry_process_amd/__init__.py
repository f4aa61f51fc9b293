"""MI355X-native log-likelihood hot path of starry_process."""
__version__ = "0.1.0"

import os as _os

# Several independent evaluations are kept in flight on separate HIP streams (engine.engine_slots,
# calibrate.EnsembleLogProb, bench.py).  The runtime maps streams onto 4 hardware queues by default: a
# fifth stream shares a queue with another and the two serialise (measured: four steps in flight
# LOSE 10 % with 4 queues and gain 3 % with 8).  Takes effect only if no HIP call has been made yet
# in this process; an explicit setting of the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .defaults import defaults  # noqa: E402,F401
from .temporal import ExpSquaredKernel, Matern32Kernel  # noqa: E402,F401


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
