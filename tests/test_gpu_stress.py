"""
Randomised parity sweep of the device log-likelihood against the oracle: sizes around the
64-column panel boundaries (1, 2, 3, 17, 63, 64, 65, ..., 400), 1-5 light curves sharing a
covariance, marginal / conditional, normalised or not, both temporal kernels, limb
darkening, scalar / per-cadence noise, baseline mean and variance.  tools/stress.py is the
stand-alone form with a printed table.
"""
import os

import numpy as np
import pytest

from conftest import golden
from oracle import sp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_configurations(seed):
    from starry_process_amd import StarryProcess
    from starry_process_amd import temporal as tmod

    rng = np.random.RandomState(seed)
    worst = 0.0
    for case in range(16):
        L = int(rng.choice([5, 15]))
        mom = golden("moments_L%d" % L)
        mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
        K = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 100, 127, 128, 129, 200, 257, 320, 400]))
        M = int(rng.choice([1, 1, 1, 2, 5]))
        marg = bool(rng.rand() < 0.6)
        normalized = bool(rng.rand() < 0.6)
        tau = None if rng.rand() < 0.6 else float(rng.uniform(0.5, 5.0))
        tk = str(rng.choice(["Matern32Kernel", "ExpSquaredKernel"]))
        u = [0.0, 0.0] if rng.rand() < 0.5 else list(rng.uniform(0, 0.4, 2))
        p = float(rng.uniform(0.3, 3.0))
        inc = float(rng.uniform(5, 90))
        t = np.sort(rng.uniform(0, 6, K)) if rng.rand() < 0.5 else np.linspace(0, 4, K)
        flux = 1e-2 * np.sin(2 * np.pi * t / p)[None, :] * rng.rand(M, 1) + 1e-3 * rng.randn(M, K)
        data_cov = 1e-6 if rng.rand() < 0.5 else 1e-6 * (1 + rng.rand(K))
        bvar = float(rng.choice([0.0, 1e-6, 1e-2]))
        bmean = float(rng.choice([0.0, 1e-3]))
        kw = dict(marginalize_over_inclination=marg, normalized=normalized)
        okw = dict(kw)
        if tau is not None:
            kw.update(tau=tau, temporal_kernel=getattr(tmod, tk))
            okw.update(tau=tau, temporal_kernel=getattr(orc, tk))
        sp = StarryProcess(ydeg=L, mean_ylm=mu, cov_ylm=Sig, **kw)
        o = orc.OracleProcess(mu, Sig, ydeg=L, **okw)
        fl = flux[0] if M == 1 else flux
        args = dict(i=inc, p=p, u=u, baseline_mean=bmean, baseline_var=bvar)
        v = float(sp.log_likelihood(t, fl, data_cov, **args))
        r = float(o.log_likelihood(t, fl, data_cov, **args))
        assert np.isfinite(v) == np.isfinite(r), (seed, case, v, r)
        if np.isfinite(r):
            worst = max(worst, abs(v - r) / max(1.0, abs(r)))
    assert worst < 1e-8, worst
