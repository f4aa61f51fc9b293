#!/usr/bin/env python
"""Where the executed fp64 MFMA work of a cfg3 step goes, from counters.

    python tools/mfma_budget.py <pmc dir of SQ_INSTS_VALU_MFMA_MOPS_F64> <steps profiled> [S] [K]

The directory is a `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64` run of
`bench.py --no-cpu --no-extras --in-flight 1 --steps N` (one step at a time).  One MOPS unit is 512 flop (rocprofv3:
MfmaFlopsF64 = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512; a v_mfma_f64_16x16x4_f64 is 2 048 flop = 4 units).  Per kernel and
step: executed flop against the ALGORITHMIC count on K cadences, and the analytic split of the difference -- every term
a closed form of the launch structure of csrc/sp_cholesky.hip (cholesky_panel2), evaluated below."""
import collections
import csv
import glob
import os
import sys

d, steps = sys.argv[1], int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
K = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
M = 1
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
mops = collections.defaultdict(float)
ndisp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU_MFMA_MOPS_F64":
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    mops[k] += float(r["Counter_Value"])
    ndisp[k].add(r["Dispatch_Id"])

Kp = (K + M + 2 + 63) // 64 * 64
nt, nsteps, w = Kp // 64, (K + 63) // 64, 8
nact = lambda j: min(64, K - 64 * j)                                    # noqa: E731
alg_chol = S * (K ** 3 / 3.0 + 2.0 * M * K * K)                         # SURVEY 8d without the assembly's 20 K^2
alg_step = S * (K ** 3 / 3.0 + 2.0 * M * K * K + 20.0 * K * K)

# --- analytic counts of the launch structure (per step, S stars) ---------------------------------------------
alg_panel = pad_panel = inv_dup = ldinv = 0.0
for s0 in range(0, nsteps, w):
    last = min(s0 + w, nsteps - 1)
    for q in range(w):
        j = s0 + q
        if j >= nsteps:
            break
        rows_p = (nt - j - 1) * 64.0
        rows_k = max(0.0, K - (j + 1) * 64.0) + (M if rows_p > 0 else 0)
        na = nact(j)
        eag_k = sum(nact(i) ** 2 * na for i in range(j + 1, last + 1) if i < nsteps)
        eag_p = max(0, last - j) * 64.0 ** 3
        a = 2 * rows_k * na * (q * 64.0) + rows_k * na * na + eag_k + na ** 3 / 3.0
        p = 2 * rows_p * 64 * (q * 64.0) + rows_p * 64 * 64 + eag_p + 64.0 ** 3 / 3.0
        alg_panel += S * a
        pad_panel += S * (p - a)
        # the solve as a product with the explicit inverse L_d^-1 on the matrix cores: its ten 16 x 16 blocks on or below
        # the diagonal, 2 rows 64^2 (10 / 16) = 1.25 rows 64^2 where forward substitution needs rows 64^2
        inv_dup += S * 0.25 * rows_p * 64 * 64
        # forming L_d^-1 of every pivot block: 64^3 / 3 (the inverse of a triangular block)
        ldinv += S * 64.0 ** 3 / 3.0
alg_syrk = pad_syrk = diag_dup = 0.0
for s0 in range(0, nsteps, w):
    cE = (s0 + w) * 64
    if cE >= K:
        continue
    n_p, n_k, kd = Kp - cE, K + M - cE, w * 64
    alg_syrk += S * n_k * (n_k + 1.0) * kd
    pad_syrk += S * (n_p * (n_p + 1.0) - n_k * (n_k + 1.0)) * kd
    # diagonal 64 x 64 tiles: ten of their sixteen 16 x 16 blocks are on or below the diagonal (SymDeal, sp_mm.h); of
    # those the four diagonal blocks are computed in full (16 x 16 where 16 x 17 / 2 entries are wanted)
    ntile = n_p // 64
    diag_dup += S * ntile * (4 * (16 * 16 - 16 * 17 / 2.0)) * 2.0 * kd

print("cfg3 step: S = %d stars, K = %d (padded system %d), M = %d; %d steps profiled" % (S, K, Kp, M, steps))
print("algorithmic (SURVEY 8d): Cholesky + solves %.4g flop, whole step %.4g" % (alg_chol, alg_step))
print()
print("%-46s %10s %14s %14s" % ("kernel (counters: executed MFMA flop per step)", "launches", "flop / step", "share"))
tot = sum(mops.values()) * 512.0 / steps
for k, v in sorted(mops.items(), key=lambda kv: -kv[1]):
    if v <= 0:
        continue
    print("%-46s %10.1f %14.4g %13.1f %%" % (k[:46], len(ndisp[k]) / float(steps), v * 512.0 / steps, 100.0 * v * 512.0 / steps / tot))
print("%-46s %10s %14.4g   = %.3f x the algorithmic Cholesky + solves" % ("all kernels", "", tot, tot / alg_chol))
print()
print("analytic split of the factorisation's launches (closed forms of cholesky_panel2's schedule, per step):")
rows = [("algorithmic on K: panel launches", alg_panel), ("algorithmic on K: trailing updates", alg_syrk),
        ("padding to %d rows / 64-wide blocks: panel launches" % Kp, pad_panel),
        ("padding: trailing updates", pad_syrk),
        ("solve by the explicit inverse (ten blocks of L_d^-1: 1.25 x substitution)", inv_dup),
        ("forming L_d^-1 of %d pivot blocks" % nsteps, ldinv),
        ("trailing update: diagonal 16 x 16 blocks in full", diag_dup)]
acc = 0.0
for name, v in rows:
    acc += v
    print("   %-74s %12.4g  %6.2f %% of executed" % (name, v, 100.0 * v / tot))
print("   %-74s %12.4g  %6.2f %%" % ("sum of the terms above", acc, 100.0 * acc / tot))
print("   %-74s %12.4g  %6.2f %%" % ("rest (look-ahead / eager duplicates, residual + normalisation rows' products,",
                                    tot - acc, 100.0 * (tot - acc) / tot))
print("   %-74s" % "      assembly's block 0, table / moment products)")
