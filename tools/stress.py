#!/usr/bin/env python
"""Randomised parity sweep: StarryProcess.log_likelihood (device path) against the oracle for
random sizes and options.  python tools/stress.py [ncases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import sp_oracle as orc
from starry_process_amd import StarryProcess

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n):
    L = int(rng.choice([5, 15]))
    mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L%d.npz" % L))
    mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
    K = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 100, 127, 128, 129, 200, 257, 320, 400]))
    M = int(rng.choice([1, 1, 1, 2, 5]))
    marg = bool(rng.rand() < 0.6)
    normalized = bool(rng.rand() < 0.6)
    tau = None if rng.rand() < 0.6 else float(rng.uniform(0.5, 5.0))
    tk = rng.choice(["Matern32Kernel", "ExpSquaredKernel"])
    u = [0.0, 0.0] if rng.rand() < 0.5 else list(rng.uniform(0, 0.4, 2))
    p = float(rng.uniform(0.3, 3.0))
    inc = float(rng.uniform(5, 90))
    t = np.sort(rng.uniform(0, 6, K)) if rng.rand() < 0.5 else np.linspace(0, 4, K)
    flux = 1e-2 * np.sin(2 * np.pi * t / p)[None, :] * rng.rand(M, 1) + 1e-3 * rng.randn(M, K)
    dc_kind = rng.choice(["scalar", "vector"])
    data_cov = 1e-6 if dc_kind == "scalar" else 1e-6 * (1 + rng.rand(K))
    bvar = float(rng.choice([0.0, 1e-6, 1e-2]))
    bmean = float(rng.choice([0.0, 1e-3]))
    from starry_process_amd import temporal as tmod
    kw = dict(marginalize_over_inclination=marg, normalized=normalized)
    if tau is not None:
        kw.update(tau=tau, temporal_kernel=getattr(tmod, tk))
    sp = StarryProcess(ydeg=L, mean_ylm=mu, cov_ylm=Sig, **kw)
    okw = dict(kw)
    if tau is not None:
        okw["temporal_kernel"] = getattr(orc, tk)
    o = orc.OracleProcess(mu, Sig, ydeg=L, **okw)
    fl = flux[0] if M == 1 else flux
    v = float(sp.log_likelihood(t, fl, data_cov, i=inc, p=p, u=u, baseline_mean=bmean, baseline_var=bvar))
    r = float(o.log_likelihood(t, fl, data_cov, i=inc, p=p, u=u, baseline_mean=bmean, baseline_var=bvar))
    if np.isfinite(r) and np.isfinite(v):
        err = abs(v - r) / max(1.0, abs(r))
    else:
        err = 0.0 if (np.isinf(r) and np.isinf(v)) or (np.isnan(r) and not np.isfinite(v)) else np.inf
    worst = max(worst, err)
    flag = "" if err < 1e-8 else "   <-- MISMATCH"
    print("case %3d L=%2d K=%3d M=%d marg=%d norm=%d tau=%s u=%s dc=%s bvar=%g: gpu %.10g oracle %.10g rel %.2e%s" % (
        case, L, K, M, marg, normalized, "%.2f/%s" % (tau, tk[:3]) if tau else "-", "ld" if u[0] else "0", dc_kind, bvar, v, r, err, flag))
print("worst relative difference: %.3e" % worst)
