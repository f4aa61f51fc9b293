"""Time of one log-likelihood gradient (starry_process_amd/grad.py) at the reference's timing-test size
(tests/test_timing.py:80-145: ydeg 15, npts 1000, both branches; its soft threshold is 0.2 s).
usage: python tools/grad_timing.py [K]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from starry_process_amd.grad import hyper_gradient, log_likelihood_with_grad  # noqa: E402
from starry_process_amd.synthetic import synthetic_star                       # noqa: E402
from starry_process_amd.engine import get_engine                              # noqa: E402
from starry_process_amd.upstream_device import ylm_moments_device             # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
st = synthetic_star(1, K)
e = get_engine(15, 2)
mu, Sig = [x.cpu().numpy() for x in ylm_moments_device(e)]
for marg in (True, False):
    for what, fn in (("d/d(mu_y, Sigma_y, p)", lambda: log_likelihood_with_grad(
                          mu, Sig, st["t"], st["flux"], st["data_cov"], p=st["p"], i=st["i"],
                          marginalize_over_inclination=marg)),
                     ("d/d(r, a, b, c, n, p)", lambda: hyper_gradient(
                         st["t"], st["flux"], st["data_cov"], p=st["p"], i=st["i"], marginalize_over_inclination=marg))):
        fn()
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn()
        dt = (time.perf_counter() - t0) / 5
        print("K=%d %s %-24s %.1f ms   lnL %.6f" % (K, "marginal   " if marg else "conditional", what, dt * 1e3, out[0]))
