"""Helpers shared by the measurement scripts: an engine with the golden moments, the device inputs of a
cfg3-style step, and a run."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import Engine, make_stars
from starry_process_amd.synthetic import synthetic_star

mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))


def engine():
    e = Engine(15, 2, 0)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
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
