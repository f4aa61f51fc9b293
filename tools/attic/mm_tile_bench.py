#!/usr/bin/env python
"""The product engine's tile shapes on one batched full product C -= A B^T of the cfg5 trailing update's size
(library built with -DSP_MM_TILE_PROBE; SP_MM_TILE = 0 default choice / 1, 2, 3: 128 x 64 variants / 4: 64 x 64):
python tools/mm_tile_bench.py [batch] [M] [N] [k]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from starry_process_amd.engine import get_engine  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2432
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2432
k = int(sys.argv[4]) if len(sys.argv) > 4 else 512
e = get_engine(15, 2, 0)
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(b, M, k, dtype=torch.float64, device="cuda", generator=g)
B = torch.randn(b, N, k, dtype=torch.float64, device="cuda", generator=g)
C0 = torch.randn(b, M, N, dtype=torch.float64, device="cuda", generator=g)
C = C0.clone()
e.gemm_nt_batched(A, B, C, alpha=-1.0, beta=1)
ref = C0[0] - A[0] @ B[0].T
err = float((C[0] - ref).abs().max())
for _ in range(3):
    e.gemm_nt_batched(A, B, C, alpha=-1.0, beta=1)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
n = 10
for _ in range(n):
    e.gemm_nt_batched(A, B, C, alpha=-1.0, beta=1)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / n
tf = 2.0 * b * M * N * k / ms / 1e9
print("SP_MM_TILE=%s  b %d M %d N %d k %d: %.3f ms  %.1f TFLOP/s = %.3f of peak   max err %.2e"
      % (os.environ.get("SP_MM_TILE", "0"), b, M, N, k, ms, tf, tf / 78.6, err))
