#!/usr/bin/env python
"""Localise a Cholesky bug: compare sp_cho_factor against numpy for a few sizes.
python tools/debug_chol.py"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import get_engine

e = get_engine(15, 2, 0)
rng = np.random.RandomState(0)
for K in (16, 40, 64, 100, 128, 200, 256, 300, 520):
    X = rng.randn(K, K + 5)
    A = X @ X.T / K + np.eye(K)
    L, info = e.cho_factor(A)
    L = L.cpu().numpy()
    Lr = np.linalg.cholesky(A)
    err = np.abs(L - Lr)
    bad = np.argwhere(~(err < 1e-9))
    print("K=%4d info=%d maxerr=%.3e nbad=%d" % (K, int(info[0]), np.nanmax(err) if np.isfinite(err).any() else np.nan, len(bad)),
          "first bad:", bad[:6].tolist())
    if len(bad) and K <= 128:
        i, j = bad[0]
        print("   got", L[i, max(0, j - 2):j + 3], "want", Lr[i, max(0, j - 2):j + 3])
        print("   bad rows:", sorted(set(bad[:, 0].tolist()))[:20], "bad cols:", sorted(set(bad[:, 1].tolist()))[:20])
