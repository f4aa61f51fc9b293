"""Single light curves beyond the sizes of BASELINE.json (K = 9,000 ... 20,000) against the oracle:
the path has no size limit other than the workspace."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from tests.conftest import golden
from oracle import sp_oracle as orc
from starry_process_amd import StarryProcess
mom = golden("moments_L15")
mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
for K, tau in ((9000, None), (12000, 3.0), (20000, None)):
    rng = np.random.RandomState(K)
    t = np.sort(rng.uniform(0, 40, K))
    flux = 1e-2 * np.sin(2 * np.pi * t / 1.7) + 1e-3 * rng.randn(K)
    kw = {}
    okw = {}
    if tau:
        from starry_process_amd import temporal as tm
        kw = dict(tau=tau, temporal_kernel=tm.Matern32Kernel); okw = dict(tau=tau, temporal_kernel=orc.Matern32Kernel)
    sp = StarryProcess(ydeg=15, mean_ylm=mu, cov_ylm=Sig, **kw)
    t0 = time.time(); v = float(sp.log_likelihood(t, flux, 1e-6, p=1.7)); t1 = time.time()
    o = orc.OracleProcess(mu, Sig, ydeg=15, **okw)
    r = float(o.log_likelihood(t, flux, 1e-6, p=1.7)); t2 = time.time()
    print(K, v, r, abs(v - r) / abs(r), "gpu %.2fs cpu %.2fs" % (t1 - t0, t2 - t1), flush=True)
