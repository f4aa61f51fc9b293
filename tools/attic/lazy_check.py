#!/usr/bin/env python
"""Covariance tiles formed at first touch (sp_set_lazy_cov) against the materialised assembly:
same handle mode (one launch per panel), a sweep of sizes and options -- the values must be
IDENTICAL -- and the time of a K = 1000, 64-star step one at a time.
python tools/lazy_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from starry_process_amd.engine import make_stars
from starry_process_amd.synthetic import synthetic_star
from _common import engine

def values(e, S, K, M=1, tau=None, nobs=None, u=(0.0, 0.0)):
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    fl = np.array([[np.roll(s["flux"], 7 * m) * (1.0 + 0.01 * m) for m in range(M)] for s in sts])
    stars = make_stars(S, period=[s["p"] for s in sts], data_var=1e-6, tau=tau or 0.0,
                       nobs=nobs if nobs is not None else 0)
    tab, mv = e.kernel_table(e.f64(e.rTA1L(list(u))), 300)
    out, status = e.lnlike_ensemble(t_d, e.f64(fl), e.stars_to_device(stars), tab=tab, meanvar=mv,
                                    temporal="matern32" if tau else None)
    torch.cuda.synchronize()
    return out.cpu().numpy().copy(), status.cpu().numpy().copy()

el, em = engine(), engine()
el.set_lazy_cov(True); em.set_lazy_cov(False)
cases = [dict(S=8, K=200), dict(S=9, K=127), dict(S=5, K=128), dict(S=3, K=129), dict(S=16, K=513),
         dict(S=8, K=960, M=70), dict(S=64, K=1000), dict(S=7, K=1345), dict(S=12, K=1100, M=5),
         dict(S=1, K=700), dict(S=6, K=1000, tau=2.5), dict(S=6, K=640, nobs=[640, 639, 500, 130, 65, 3]),
         dict(S=4, K=1000, u=(0.4, 0.2))]
for kw in cases:
    a, sa = values(el, **kw)
    b, sb = values(em, **kw)
    print("%-60s identical %s  max |rel diff| %.1e  status %d/%d finite %s" % (
        kw, bool(np.array_equal(a, b)), float(np.max(np.abs(a / b - 1))), int(sa.any()), int(sb.any()),
        bool(np.all(np.isfinite(a)))), flush=True)
for name, e in (("materialised", em), ("first touch", el)):
    sts = [synthetic_star(s, 1000) for s in range(64)]
    t_d = e.f64(np.array([s["t"] for s in sts])); f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
    st_d = e.stars_to_device(make_stars(64, period=[s["p"] for s in sts], data_var=1e-6))
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    ws = e.workspace(64, 1000, 1); out = e.empty(64)
    for _ in range(30):
        e.lnlike_ensemble(t_d, f_d, st_d, tab=tab, meanvar=mv, out=out, workspace=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        e.lnlike_ensemble(t_d, f_d, st_d, tab=tab, meanvar=mv, out=out, workspace=ws)
    torch.cuda.synchronize()
    print("%-14s %.4f ms per 64-star step (one launch per panel, one step at a time)" % (name, (time.perf_counter() - t0) / 200 * 1e3))
