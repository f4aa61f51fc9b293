#!/usr/bin/env python
"""Secondary measurement: BASELINE cfg5 shape on ONE GPU (ydeg = 20, K = 3000, t = linspace(0, 30),
tau = 3 Matern-3/2, u = [0.4, 0.2]; cfg5 proper is 256 stars over 8 GPUs = 32 per GPU).
python tools/bench_cfg5.py [stars]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import get_engine, make_stars
from starry_process_amd.synthetic import synthetic_star

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, L = 3000, 20
e = get_engine(L, 2, 0)
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L20.npz"))
mu_d, Sig_d = e.f64(mom["default_mean_ylm"]), e.f64(mom["default_cov_ylm"])
e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
sts = [synthetic_star(s, K, 30.0) for s in range(S)]
t_d = e.f64(np.array([s["t"] for s in sts])); f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
stars_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], tau=3.0, data_var=1e-6))
rta1 = e.f64(e.rTA1L([0.4, 0.2]))
ws = e.workspace(S, K, 1)
out = e.empty(S); status = torch.zeros(S, dtype=torch.int32, device=e.device)
def step():
    e.set_moments_dev(mu_d, Sig_d)
    tab, mv = e.kernel_table(rta1, 300)
    e.lnlike_ensemble(t_d, f_d, stars_d, covpts=300, tab=tab, meanvar=mv, temporal="matern32",
                      normalized=True, out=out, status=status, workspace=ws)
for _ in range(2): step()
steps = 5
e.profile_begin(steps * 64)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
launches, ms, flops = e.profile_end()
F = K ** 3 / 3 + 2 * K ** 2 + 20 * K ** 2
res = {"workload": "cfg5 shape: ydeg=20, K=3000, tau=3 matern32, u=[0.4,0.2], %d stars on 1 GPU" % S,
       "evals_per_s": S * steps / dt, "ms_per_step": 1e3 * dt / steps,
       "algorithmic_TFLOPs_whole_step": S * steps * F / dt * 1e-12,
       "trailing_update_TFLOPs": flops / (ms * 1e-3) * 1e-12 if ms > 0 else None,
       "trailing_update_frac_of_78.6": flops / (ms * 1e-3) * 1e-12 / 78.6 if ms > 0 else None,
       "finite": bool(np.isfinite(out.cpu().numpy()).all()), "status_any": bool(status.cpu().numpy().any())}

# the same with three independent steps in flight (engine_slots; bench.py --in-flight)
from starry_process_amd.engine import engine_slots
slots = []
for ek, stream in engine_slots(L, 2, 0, 3):
    ek.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    slots.append((ek, stream, ek.workspace(S, K, 1), ek.empty(S)))
torch.cuda.synchronize()
def step_on(c):
    ek, stream, wsk, outk = c
    with torch.cuda.stream(stream):
        ek.set_moments_dev(mu_d, Sig_d)
        tab, mv = ek.kernel_table(rta1, 300)
        ek.lnlike_ensemble(t_d, f_d, stars_d, covpts=300, tab=tab, meanvar=mv, temporal="matern32",
                           normalized=True, out=outk, workspace=wsk)
for i in range(6): step_on(slots[i % 3])
steps3 = 15
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps3): step_on(slots[i % 3])
torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
res["three_steps_in_flight"] = {"evals_per_s": S * steps3 / dt3, "ms_per_step": 1e3 * dt3 / steps3,
                                "algorithmic_TFLOPs_whole_step": S * steps3 * F / dt3 * 1e-12,
                                "same_bits": bool(torch.equal(slots[1][3], out) and torch.equal(slots[2][3], out))}
print(json.dumps(res))
