"""Where the host time of the batched-samples steps goes WITH the steps in flight: cumulative host time of each call
of SampleBatches' loop (no synchronisation until the end)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from starry_process_amd.calibrate import SampleBatches  # noqa: E402
from starry_process_amd.engine import engine_slots, make_stars, sample_parameters  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402

Sd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = 1000
sts = [synthetic_star(s, K) for s in range(Sd)]
slots = engine_slots(15, 2, 0, F)
e0 = slots[0][0]
stars = make_stars(Sd, period=[s["p"] for s in sts], data_var=1e-6)
sb = SampleBatches(slots, e0.f64(np.array([s["t"] for s in sts])), e0.f64(np.array([s["flux"] for s in sts])[:, None, :]),
                   stars, e0.f64(e0.rTA1L([0.0, 0.0])), 300)
g = sb.group
rng = np.random.RandomState(7)
smp = np.column_stack([rng.uniform(15.0, 25.0, g), rng.uniform(0.3, 0.5, g), rng.uniform(0.2, 0.35, g),
                       rng.uniform(0.08, 0.12, g), rng.uniform(5.0, 12.0, g)])
acc = {}


def timed(name, fn):
    t0 = time.perf_counter()
    fn()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


steps = 48
for rnd in range(2):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        (e, stream), b = sb._slots[it % F], sb._buf[it % F]
        with torch.cuda.stream(stream):
            timed("polar_moments_samples", lambda: e.polar_moments_samples(smp, ez=b["ez"], Ez=b["Ez"]))
            timed("kernel_table_samples", lambda: e.kernel_table_samples(b["ez"], b["Ez"], sb._rta1, 300, tab=b["tab"], meanvar=b["mv"]))
            timed("lnlike_planned", lambda: e.lnlike_ensemble_planned(sb._plan, None, None, sb._stars, b["tab"], b["mv"], workspace=b["ws"]))
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
print("Sd %d F %d: host %.3f ms per step, total %.3f ms per step" % (Sd, F, 1e3 * host / steps, 1e3 * total / steps))
for k, v in acc.items():
    print("%-24s %8.1f us per step" % (k, 1e6 * v / steps))
