"""The small-K kernel (csrc/sp_small.hip) alone: time of the planned call against the number of stars -- one workgroup
per star, so S = 256 is one workgroup per CU (its latency alone), 512 / 768 two / three per CU, beyond that rounds.
python tools/small_k_probe.py [K ...]        (SP_PROBE_S=256,3072 picks the star counts)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from starry_process_amd.engine import Engine, make_stars  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402

e = Engine(15, 2, 0)
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
for K in [int(x) for x in sys.argv[1:]] or [64, 128]:
    base = [synthetic_star(s, K) for s in range(64)]
    for S in [int(x) for x in os.environ.get("SP_PROBE_S", "8,256,512,768,1536,3072").split(",")]:
        sts = [base[s % 64] for s in range(S)]
        t_d = e.f64(np.array([s["t"] for s in sts]))
        f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
        s_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
        plan = e.plan_data(t_d, f_d, s_d, covpts=300)
        out = e.empty(S)
        st = torch.zeros(S, dtype=torch.int32, device=e.device)
        for _ in range(5):
            e.lnlike_ensemble_planned(plan, None, None, s_d, tab, mv, out=out, status=st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        a.record()
        for _ in range(n):
            e.lnlike_ensemble_planned(plan, None, None, s_d, tab, mv, out=out, status=st)
        b.record()
        torch.cuda.synchronize()
        print("K %4d  S %5d   %8.1f us per call   %6.2f us x CU-slot per star" % (K, S, 1e3 * a.elapsed_time(b) / n,
                                                                           1e3 * a.elapsed_time(b) / n / max(S / 256.0, 1.0)))
