#!/usr/bin/env python
"""How do the kernels of a step scale with the number of streams they run on?  F independent
streams (own handle, own data) each running ONE kind of work back to back:
    syrk   the rank-512 symmetric trailing update of 64 x (512 x 512) blocks (mm_nt_kernel)
    chol   the whole factorisation of 64 x (1024 x 1024) systems (sp_cho_factor, panel mode as set)
    step   the whole likelihood step
Prints the aggregate rate per F.  python tools/inflight_probe.py [F ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from starry_process_amd._lib import check
from _common import engine, setup, run

Fs = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]
S, K = 64, 1000
torch.manual_seed(0)


def spd(e, S, n):
    A = torch.randn(S, n, n, dtype=torch.float64, device=e.device) * 0.01
    return A @ A.transpose(1, 2) + torch.eye(n, dtype=torch.float64, device=e.device)


for F in Fs:
    slots = []
    for k in range(F):
        e = engine()
        st = torch.cuda.Stream(device=e.device)
        X = torch.randn(S, 512, 512, dtype=torch.float64, device=e.device)
        C = torch.zeros(S, 512, 512, dtype=torch.float64, device=e.device)
        A0 = spd(e, S, 1024)
        a = setup(e, S, K)
        slots.append((e, st, X, C, A0, A0.clone(), a))
    torch.cuda.synchronize()

    def timed(fn, reps):
        for e, st, *r in slots:
            with torch.cuda.stream(st):
                for _ in range(3):
                    fn(e, st, *r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for e, st, *r in slots:
                with torch.cuda.stream(st):
                    fn(e, st, *r)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps / F

    def syrk(e, st, X, C, A0, A1, a):
        e.gemm_nt_batched(X, X, C, alpha=-1.0, beta=1, lower_only=True)

    def chol(e, st, X, C, A0, A1, a):
        A1.copy_(A0)
        info = torch.zeros(S, dtype=torch.int32, device=e.device)
        check(e._L.sp_cho_factor(e._h, e._p(A1), 1024, 1024, 1024 * 1024, S, e._p(info), st.cuda_stream))

    def copy_only(e, st, X, C, A0, A1, a):
        A1.copy_(A0)
        info = torch.zeros(S, dtype=torch.int32, device=e.device)

    def step(e, st, X, C, A0, A1, a):
        t_d, f_d, stars_d, tab, mv = a
        e.lnlike_ensemble(t_d, f_d, stars_d, tab=tab, meanvar=mv)

    ts = timed(syrk, 40)
    fl = S * 512 * 513 * 512.0
    tcp = timed(copy_only, 40)
    tc = timed(chol, 40)
    tst = timed(step, 40)
    print("F=%d  syrk %.1f us/launch-equivalent (%.1f TFLOP/s aggregate) | chol %.3f ms per 64 systems (copy %.3f; %.1f TFLOP/s of n^3/3 net of the copy) | step %.3f ms (%.0f evals/s)" % (
        F, 1e6 * ts, fl / ts / 1e12, 1e3 * tc, 1e3 * tcp, S * 1024 ** 3 / 3.0 / max(tc - tcp, 1e-9) / 1e12, 1e3 * tst, S / tst), flush=True)
    del slots
