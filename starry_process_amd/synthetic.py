"""
Deterministic synthetic light curves used by the tests, the golden-vector
generator and ``bench.py`` ("Synthetic inputs").

Star ``s`` (0-based)::

    rng   = np.random.RandomState(1000 + s)
    t     = linspace(0, tspan, K)
    p_s   = 1.0 if s == 0 else rng.uniform(0.5, 2.0)
    i_s   = arccos(rng.uniform(0, 1)) * 180 / pi        (degrees)
    flux  = 1e-2 * sin(2 pi t / p_s) + 1e-3 * rng.randn(K)
    data_cov = 1e-6

The draw order (period, inclination, noise) is part of the definition.
"""
import numpy as np

__all__ = ["synthetic_star", "synthetic_ensemble"]


def synthetic_star(s, K, tspan=4.0):
    rng = np.random.RandomState(1000 + int(s))
    t = np.linspace(0.0, tspan, K)
    p = rng.uniform(0.5, 2.0)
    inc = np.arccos(rng.uniform(0.0, 1.0)) * 180.0 / np.pi
    if s == 0:
        p = 1.0
    flux = 1e-2 * np.sin(2 * np.pi * t / p) + 1e-3 * rng.randn(K)
    return dict(t=t, flux=flux, p=float(p), i=float(inc), data_cov=1e-6)


def synthetic_ensemble(first, count, K, tspan=4.0):
    """Stars ``first .. first+count-1`` stacked: t (K,), flux (S,K), p (S,),
    i (S,), data_cov (S,)."""
    stars = [synthetic_star(s, K, tspan) for s in range(first, first + count)]
    return dict(
        t=stars[0]["t"] if count else np.linspace(0.0, tspan, K),
        flux=np.array([st["flux"] for st in stars]).reshape(count, K),
        p=np.array([st["p"] for st in stars]),
        i=np.array([st["i"] for st in stars]),
        data_cov=np.array([st["data_cov"] for st in stars]),
    )
