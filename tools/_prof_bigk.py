import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.conftest import golden
from starry_process_amd import StarryProcess
mom = golden("moments_L15")
mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
K = 8000
rng = np.random.RandomState(K)
t = np.sort(rng.uniform(0, 40, K))
flux = 1e-2 * np.sin(2 * np.pi * t / 1.7) + 1e-3 * rng.randn(K)
sp = StarryProcess(ydeg=15, mean_ylm=mu, cov_ylm=Sig)
for it in range(3):
    v = float(sp.log_likelihood(t, flux, 1e-6, p=1.7))
torch.cuda.synchronize()
