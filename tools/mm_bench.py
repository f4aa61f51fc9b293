#!/usr/bin/env python
"""Time the batched fp64 NT product (sp_gemm_nt) in the shapes the factorisation uses, for every
tile shape of the pipelined kernel (sp_debug_set_mm_variant), with a check against torch.

    python tools/mm_bench.py [variants...]          # default: 0 1 2 3 5 6 7 8

Shapes (64 matrices each): the rank-512 trailing update of K = 1000 (n = 512, lower tiles only),
a rank-256 update (n = 768), the conditional-covariance products, a block-column product."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch

from starry_process_amd._lib import check
from starry_process_amd.engine import get_engine

PEAK = 78.6
e = get_engine(15, 2, 0)
L = e._L
st = e._stream()
gen = torch.Generator(device="cuda").manual_seed(1)


def product(A, B, C, alpha, beta, lower):
    """A [b, M, K], B [b, N, K], C [b, M, N] (in place)."""
    b, M, K = A.shape
    N = B.shape[1]
    check(L.sp_gemm_nt(e._h, e._p(A), A.stride(1), A.stride(0), e._p(B), B.stride(1), B.stride(0),
                       e._p(C), C.stride(1), C.stride(0), M, N, K, float(alpha), int(beta),
                       int(lower), b, st))


def run(name, b, M, N, K, lower, alpha=-1.0, beta=1, same=False, reps=20, variants=(0, 1)):
    # operands live inside a wider matrix, like the panels of a padded system
    ld = max(M, N, K) + 64
    big = torch.rand(b, max(M, N), ld, generator=gen, device="cuda", dtype=torch.float64) - 0.5
    A = big[:, :M, :K]
    B = A if same else (torch.rand(b, N, ld, generator=gen, device="cuda", dtype=torch.float64) - 0.5)[:, :, :K]
    C0 = torch.rand(b, M, N, generator=gen, device="cuda", dtype=torch.float64)
    ref = (C0 if beta else 0) + alpha * torch.bmm(A, B.transpose(1, 2))
    flops = 2.0 * b * M * N * K * (0.5 * (M + 1) / M if lower else 1.0)
    out = []
    for v in variants:
        check(L.sp_debug_set_mm_variant(int(v)))
        C = C0.clone()
        product(A, B, C, alpha, beta, lower)
        torch.cuda.synchronize()
        if lower:
            err = (torch.tril(C - ref)).abs().max().item()
            keep = (torch.triu(C - C0, 64)).abs().max().item()   # tiles above the diagonal untouched
        else:
            err, keep = (C - ref).abs().max().item(), 0.0
        for _ in range(3):
            product(A, B, C, alpha, beta, lower)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            product(A, B, C, alpha, beta, lower)
        z.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(z) / reps * 1e3
        tf = flops / us * 1e-6
        out.append((v, us, tf, err, keep))
        print("%-28s variant %d: %8.1f us  %6.1f TFLOP/s (%.2f of peak)  max err %.1e%s" % (
            name, v, us, tf, tf / PEAK, err, "" if keep == 0.0 else "  UPPER TILES TOUCHED %.1e" % keep), flush=True)
    return out


if __name__ == "__main__":
    variants = [int(x) for x in sys.argv[1:]] or [0, 6, 8, 9, 10, 11, 12]
    run("syrk n=512 k=512 lower", 64, 512, 512, 512, True, same=True, variants=variants)
    run("syrk n=768 k=256 lower", 64, 768, 768, 256, True, same=True, variants=variants)
    run("syrk n=256 k=256 lower", 64, 256, 256, 256, True, same=True, variants=variants)
    run("gemm 512x512 k=512", 64, 512, 512, 512, False, variants=variants)
    run("blockcol 960x64 k=448", 64, 960, 64, 448, False, variants=[v for v in variants if v in (0, 1, 2, 6, 7, 10, 11)])
    run("cond A Sigma (1024x256x256)", 16, 1024, 256, 256, False, alpha=1.0, beta=0, variants=variants)
    run("cond B A^T (1024x1024x256)", 16, 1024, 1024, 256, False, alpha=1.0, beta=0, variants=variants)
    run("alpha=0.5 beta=1 general", 8, 256, 256, 128, False, alpha=0.5, beta=1, variants=variants)
    check(L.sp_debug_set_mm_variant(1))
