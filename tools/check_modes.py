#!/usr/bin/env python
"""Log-likelihoods of the cfg3 batch under every driver / panel mode / product variant, several
runs each: all must agree to rounding and repeat bit for bit.
python tools/check_modes.py [S] [K]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import Engine, make_stars
from starry_process_amd.synthetic import synthetic_star
from starry_process_amd._lib import check

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
sts = [synthetic_star(s, K) for s in range(S)]
ref = None
for chol in (2, 0):
    for panel in (0, 1):
        for mmv in (11, 6, 0):
            e = Engine(15, 2, 0)
            e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
            check(e._L.sp_set_chol_mode(e._h, chol))
            e.set_panel_mode(bool(panel))
            check(e._L.sp_debug_set_mm_variant(mmv))
            t_d = e.f64(np.array([s["t"] for s in sts])); f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
            stars_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
            tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
            outs = []
            for rep in range(4):
                out, status = e.lnlike_ensemble(t_d, f_d, stars_d, tab=tab, meanvar=mv)
                torch.cuda.synchronize()
                outs.append(out.cpu().numpy().copy())
                st = status.cpu().numpy()
            if ref is None:
                ref = outs[0]
            same = all(np.array_equal(outs[0], o) for o in outs[1:])
            err = np.max(np.abs(outs[0] / ref - 1))
            bad = [int(np.sum(o != outs[0])) for o in outs[1:]]
            print("chol %d panel %d mm %2d: repeats %s (differing stars %s)  status nonzero %d  max rel diff vs first config %.2e  finite %s"
                  % (chol, panel, mmv, same, bad, int(np.count_nonzero(st)), err, bool(np.all(np.isfinite(outs[0])))), flush=True)
check(e._L.sp_debug_set_mm_variant(11))
