#!/usr/bin/env python
"""Reference point for the roofline: what the vendor fp64 GEMM (rocBLAS / hipBLASLt through
torch) reaches on this device, for a large square product and for the shape of the trailing
update (64 x [768 x 256] . [256 x 768])."""
import torch, time
def bench(f, flops, name, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("%-44s %9.3f ms  %6.1f TFLOP/s" % (name, ms, flops / ms * 1e-9))
dev = "cuda"
for n in (4096, 8192):
    A = torch.randn(n, n, dtype=torch.float64, device=dev); B = torch.randn(n, n, dtype=torch.float64, device=dev)
    bench(lambda: A @ B, 2.0 * n ** 3, "dgemm %d^3" % n)
X = torch.randn(64, 768, 256, dtype=torch.float64, device=dev)
C = torch.randn(64, 768, 768, dtype=torch.float64, device=dev)
bench(lambda: torch.baddbmm(C, X, X.transpose(1, 2), beta=1.0, alpha=-1.0), 64 * 2.0 * 768 * 768 * 256, "batched 64 x (768x256)(256x768) full square")
X = torch.randn(64, 1024, 1024, dtype=torch.float64, device=dev) 
bench(lambda: torch.linalg.cholesky(X @ X.transpose(1, 2) + 1024 * torch.eye(1024, dtype=torch.float64, device=dev)), 1, "X X^T + cholesky 64 x 1024 (ms only)", reps=3)
S = X @ X.transpose(1, 2) + 1024 * torch.eye(1024, dtype=torch.float64, device=dev)
bench(lambda: torch.linalg.cholesky(S), 64 * 1024 ** 3 / 3, "torch.linalg.cholesky 64 x 1024^2", reps=3)
