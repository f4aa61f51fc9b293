import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from starry_process_amd.grad import EnsembleGradient, hyper_gradient
from starry_process_amd.synthetic import synthetic_star
# small ydeg 20 check against the single-star route
S, K = 3, 200
sts = [synthetic_star(s, K, 30.0) for s in range(S)]
t, f, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
eg = EnsembleGradient(t, f, ferr=1e-3, p=p, ydeg=20, tau=3.0, u=[0.4, 0.2])
tot, g = eg(r=20.0, a=0.4, b=0.27, c=0.1, n=10.0)
ref = {k: 0.0 for k in g}; rl = 0.0
for s in range(S):
    l1, g1 = hyper_gradient(t[s], f[s], 1e-6, p=float(p[s]), ydeg=20, tau=3.0, u=np.array([0.4, 0.2]), r=20.0, a=0.4, b=0.27, c=0.1, n=10.0)
    rl += l1
    for k in ref: ref[k] += g1[k]
print("ydeg 20, K 200, Matern: lnL", tot, rl, " max rel grad err", max(abs(g[k] - ref[k]) / max(abs(ref[k]), 1e-3 * max(abs(v) for v in ref.values())) for k in g))
# cfg5's shape: 32 stars, K = 3000
S, K = 32, 3000
sts = [synthetic_star(s, K, 30.0) for s in range(S)]
t, f, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
eg = EnsembleGradient(t, f, ferr=1e-3, p=p, ydeg=20, tau=3.0, u=[0.4, 0.2])
for _ in range(2): tot, g = eg()
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(3): tot, g = eg(r=20.0 + 0.01 * k)
torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 3
print("cfg5 shape (32 stars, K 3000, ydeg 20, Matern): %.1f ms per gradient; lnL %.3f; finite %s" % (ms, tot, all(np.isfinite(v) for v in g.values())), g)
