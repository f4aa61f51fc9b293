#!/usr/bin/env python
"""The dataflow panel chain (sp_set_chol_mode 3) against the super-panel driver: values on a sweep
of sizes, then the time of a K = 1000, 64-star step one at a time under each driver.
python tools/chain_check.py [quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import Engine, make_stars
from starry_process_amd.synthetic import synthetic_star
from starry_process_amd._lib import check

mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))


def engine(chol, panel=0):
    e = Engine(15, 2, 0)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    check(e._L.sp_set_chol_mode(e._h, chol))
    e.set_panel_mode(bool(panel))
    return e


def setup(e, S, K, M=1):
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    if M == 1:
        fl = np.array([s["flux"] for s in sts])[:, None, :]
    else:
        fl = np.array([[np.roll(s["flux"], 7 * m) * (1.0 + 0.01 * m) for m in range(M)] for s in sts])
    f_d = e.f64(fl)
    stars_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    return t_d, f_d, stars_d, tab, mv


def run(e, args, reps=1):
    t_d, f_d, stars_d, tab, mv = args
    outs = []
    for _ in range(reps):
        out, status = e.lnlike_ensemble(t_d, f_d, stars_d, tab=tab, meanvar=mv)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().copy())
    return outs, status.cpu().numpy()


def main():
    e0, e3 = engine(0, 1), engine(3)
    cases = [(8, 200, 1), (9, 40, 1), (5, 64, 1), (3, 65, 1), (16, 513, 1), (8, 960, 70), (64, 1000, 1), (7, 1345, 1),
             (12, 1100, 5), (1, 700, 1)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cases = cases[:2]
    for S, K, M in cases:
        a0 = setup(e0, S, K, M)
        ref, st0 = run(e0, a0)
        t0 = time.time()
        outs, st3 = run(e3, setup(e3, S, K, M), reps=3)
        dt = time.time() - t0
        scale = np.maximum(np.abs(ref[0]), np.abs(ref[0]).max())
        err = np.max(np.abs(outs[0] - ref[0]) / scale)
        same = all(np.array_equal(outs[0], o) for o in outs[1:])
        print("S %3d K %5d M %2d: max rel diff %.2e  repeats %s  status %d/%d  finite %s  (%.2f s)"
              % (S, K, M, err, same, int(np.count_nonzero(st0)), int(np.count_nonzero(st3)),
                 bool(np.all(np.isfinite(outs[0]))), dt), flush=True)

    # time per step, one at a time
    for name, e in (("super-panel, two launches", engine(0, 0)), ("super-panel, one launch", e0), ("dataflow chain", e3)):
        a = setup(e, 64, 1000)
        for w in (0,):
            run(e, a, reps=30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                e.lnlike_ensemble(a[0], a[1], a[2], tab=a[3], meanvar=a[4])
            torch.cuda.synchronize()
            print("%-28s %.4f ms per 64-star step" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)


if __name__ == "__main__":
    main()
