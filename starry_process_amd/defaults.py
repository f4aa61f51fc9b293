"""
Default hyperparameters and numerical settings, grouped by what consumes them.  The VALUES
are the reference's (``defaults.py:4-35``) -- they are part of the interface: scripts read
``defaults["r"]``, the constructor keywords fall back on them, and the fixtures under
``tests/golden`` were generated with them.
"""
import numpy as np

from .temporal import Matern32Kernel

# expansion degrees: spherical harmonics of the surface map, limb-darkening polynomial
_degrees = {"ydeg": 15, "udeg": 2}

# the spot population (constructor keywords of StarryProcess)
_spots = {
    "r": 20.0,      # mean angular radius, degrees
    "dr": None,     # half-width of a uniform radius distribution (None: delta function)
    "a": 0.40,      # latitude distribution, Beta shape parameters mapped to the unit square
    "b": 0.27,
    "c": 0.1,       # contrast
    "n": 10.0,      # number of spots
}

# the star and the observation (keywords of flux / cov / log_likelihood)
_star = {
    "i": 60.0,                  # inclination, degrees
    "p": 1.0,                   # rotation period, units of t
    "u": np.zeros(30),          # limb-darkening coefficients; the first udeg are used
    "baseline_mean": 0.0,
    "baseline_var": 0.0,
}

# structure of the Gaussian process
_process = {
    "tau": None,                            # time scale of the temporal kernel (None: static)
    "temporal_kernel": Matern32Kernel,
    "normalized": True,                     # light curves divided by their mean
    "marginalize_over_inclination": True,
    "normalization_order": 20,              # terms of the alpha(z), beta(z) series
    "normalization_zmax": 0.023,            # beyond it the series is not trusted: -inf
    "covpts": 300,                          # lag-grid points of the marginal kernel spline
}

# numerical regularisation and the latitude-parameter transforms
_numerics = {
    "eps": 1e-8,            # flux covariance jitter
    "epsy": 1e-12,          # Ylm covariance jitter, l < 15
    "epsy15": 1e-9,         # ... l >= 15
    "driver": "numpy",      # eigensolver of matrix_sqrt
    "log_alpha_max": 10,
    "log_beta_max": 10,
    "abmin": 1e-12,
    "sigma_max": 45.0,      # widest latitude mode (degrees) the (mu, sigma) form accepts
}

# map-rendering resolution (visualisation helpers of the reference; unused on this path)
_render = {"mx": 300, "my": 150}

defaults = {}
for _group in (_degrees, _spots, _star, _process, _numerics, _render):
    defaults.update(_group)
del _group
