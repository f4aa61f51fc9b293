"""
CPU checks of what the batched sampler upstream (csrc/sp_samples.hip, round 6) relies on -- no GPU:

  * the polar-frame identity: the longitude average of the quadrature of rotations is the projection of
    M = sum_k w_k u_k^T u_k onto the matrices commuting with every Rz, so that
        ez = sqrt(n) e1,   Ez = g^2 Proj(M) + (n - 1) e1 e1^T + diag(eps)
    equals oracle.polar_moments(oracle.ylm_moments_quadrature(...)) -- the 2 (ydeg + 2) (2 ydeg + 3) rotations of the
    per-sample path (latitude.py:199-212, longitude.py:19-24, contrast.py:18-33, flux.py:54-62 of the reference);
  * the Gauss-Jacobi rule by bisection on the Sturm sequence + the orthonormal recurrence (a NumPy restatement of
    sm_prepare_kernel's arithmetic) against the library's Golub-Welsch rule over the reference's prior box, the
    symmetric case alpha = beta (zero diagonal: an exact zero pivot at the first midpoint) included;
  * the host helpers of the batch: parameter map and bounds, the stars of (sample, star) systems, the vectorised
    log-Jacobian, the stream-depth clamp.
"""
import warnings

import numpy as np
import pytest

from starry_process_amd import upstream


def projected_polar_moments(ydeg, r, a, b, c, n, gj=None):
    """(ez, Ez) the way csrc/sp_samples.hip computes them, with the oracle's CPU rotations."""
    from oracle import sp_oracle as orc

    N = (ydeg + 1) ** 2
    s1, _ = upstream.size_moments(r, None, ydeg)
    alpha, beta = upstream.ab_to_alphabeta(a, b)
    t, w = (gj or orc.gauss_jacobi)(ydeg + 2, beta - 1.0, alpha - 1.0)
    x = 0.5 * (1.0 + t)
    phis = np.concatenate([np.arccos(x), -np.arccos(x)])
    wphi = 0.5 * np.concatenate([w, w])
    Rp = orc.Rx(ydeg, 0.5 * np.pi)[0]
    tabs = orc.index_tables(ydeg)
    m_of, mirror = tabs["m_of"], tabs["mirror"]
    g = np.pi * c * np.sqrt(n)
    M, e1 = np.zeros((N, N)), np.zeros(N)
    for ph, wk in zip(phis, wphi):
        u = orc.dotRx(ydeg, orc.dotRx(ydeg, s1[None, :], orc.Rx(ydeg, ph)[0]), Rp)[0]
        M += wk * np.outer(u, u)
        e1 += g * wk * np.where(m_of == 0, u, 0.0)
    Mm = M[np.ix_(mirror, mirror)]
    same = m_of[:, None] == m_of[None, :]
    opp = (m_of[:, None] == -m_of[None, :]) & (m_of[:, None] != 0)
    G = np.where(same, 0.5 * (M + Mm), 0.0) + np.where(opp, 0.5 * (M - Mm), 0.0)
    lam = np.ones(N) * 1e-12
    lam[15 ** 2:] = 1e-9
    return np.sqrt(n) * e1, g * g * G + (n - 1.0) * np.outer(e1, e1) + np.diag(lam)


@pytest.mark.parametrize("hyper", [(20.0, 0.4, 0.27, 0.1, 10.0), (12.0, 0.9, 0.05, 0.2, 3.0), (28.0, 1.0, 1.0, 0.05, 1.0)])
def test_polar_frame_projection_is_the_quadrature_of_rotations(hyper):
    from oracle import sp_oracle as orc

    ydeg = 6
    r, a, b, c, n = hyper
    s1, _ = upstream.size_moments(r, None, ydeg)
    alpha, beta = upstream.ab_to_alphabeta(a, b)
    mu, Sig = orc.ylm_moments_quadrature(s1, s1[None, :], alpha, beta, c, n, ydeg)
    ez, Ez = orc.polar_moments(ydeg, mu, Sig)
    ez2, Ez2 = projected_polar_moments(ydeg, *hyper)
    assert np.abs(ez.ravel() - ez2).max() < 1e-13 * np.abs(ez).max()
    assert np.abs(Ez - Ez2).max() < 1e-13 * np.abs(Ez).max()


def legendre_row0(ydeg, c, s):
    """Row m' = 0 of every degree's real rotation block about x by the angle of cosine c and sine s, the way
    sm_rows_kernel computes it (normalised associated Legendre recurrence)."""
    out = np.zeros((ydeg + 1) ** 2)
    nmm = 1.0
    for m in range(ydeg + 1):
        if m > 0:
            nmm *= np.sqrt((2 * m - 1) / (2 * m)) * s
        p2, p1 = 0.0, nmm
        sg, t4 = (-1.0) ** m, m % 4
        for l in range(m, ydeg + 1):
            v = nmm
            if l > m:
                v = ((2 * l - 1) * c * p1 - np.sqrt((l - 1) ** 2 - m * m) * p2) / np.sqrt(l * l - m * m)
                p2, p1 = p1, v
            if m == 0:
                out[l * l + l] = v
            else:
                out[l * l + l + m] = sg * np.sqrt(2.0) * [1, 0, -1, 0][t4] * v
                out[l * l + l - m] = -sg * np.sqrt(2.0) * [0, 1, 0, -1][t4] * v
    return out


def test_rotated_zonal_vector_is_a_row_of_legendre_functions():
    """s Rx(phi) for a zonal s: row m' = 0 of sp_Rx's blocks (wigner.h:36-284) = d^l_{m0}(phi) in the real basis."""
    from oracle import sp_oracle as orc

    for ydeg in (5, 15, 20):
        for th in (0.3, -0.3, 1.2, -2.0, 0.5 * np.pi, 1e-3, -1e-9, 3.1):
            R = orc.Rx(ydeg, th)[0]
            ref = np.concatenate([R[(orc.nwig(l - 1) if l else 0):orc.nwig(l)].reshape(2 * l + 1, 2 * l + 1)[l]
                                  for l in range(ydeg + 1)])
            assert np.abs(legendre_row0(ydeg, np.cos(th), np.sin(th)) - ref).max() < 5e-14, (ydeg, th)


def bisection_gauss_jacobi(n, a, b):
    """sm_prepare_kernel's rule: node i = i-th eigenvalue of the Jacobi matrix by multi-section on the Sturm count (pivots
    below 1e-290 are replaced by -1e-290 BEFORE they are counted and used), weights from the orthonormal recurrence."""
    d, e = np.zeros(n), np.zeros(n)
    ab = a + b
    d[0] = (b - a) / (ab + 2.0)
    for k in range(1, n):
        s = 2.0 * k + ab
        d[k] = (b - a) * (b + a) / (s * (s + 2.0))
        num = 4.0 * (1 + a) * (1 + b) / ((s * s) * (s + 1)) if k == 1 else \
            4.0 * k * (k + a) * (k + b) * (k + ab) / ((s * s) * (s + 1) * (s - 1))
        e[k - 1] = np.sqrt(num)
    e2 = e * e

    def count_below(x):
        q = d[0] - x
        if abs(q) < 1e-290:
            q = -1e-290
        cnt = int(q < 0)
        for k in range(1, n):
            q = d[k] - x - e2[k - 1] / q
            if abs(q) < 1e-290:
                q = -1e-290
            cnt += int(q < 0)
        return cnt

    npt = min(15, 256 // n)            # trial points per node and round (a group of threads per node on the device)
    rounds, span = 0, 2.0
    while span > 3.0e-18:
        rounds, span = rounds + 1, span / (npt + 1)
    t = np.zeros(n)
    for i in range(n):
        lo, hi = -1.0, 1.0
        for _ in range(rounds):
            below = sum(1 - int(count_below(lo + (hi - lo) * ((j + 1) / (npt + 1))) > i) for j in range(npt))
            w0, l0 = hi - lo, lo
            if below > 0:
                lo = l0 + w0 * (below / (npt + 1))
            if below < npt:
                hi = l0 + w0 * ((below + 1) / (npt + 1))
        t[i] = 0.5 * (lo + hi)
    w = np.zeros(n)
    for i in range(n):
        p0, p1, s = 0.0, 1.0, 1.0
        for k in range(n - 1):
            p2 = ((t[i] - d[k]) * p1 - (e[k - 1] if k > 0 else 0.0) * p0) / e[k]
            s += p2 * p2
            p0, p1 = p1, p2
        w[i] = 1.0 / s
    return t, w / w.sum()


def test_bisection_rule_equals_golub_welsch_over_the_prior_box():
    from starry_process_amd.upstream_device import gauss_jacobi

    rng = np.random.RandomState(3)
    ab = [(0.0, 0.0), (1.0, 1.0), (0.0, 1.0), (1.0, 0.0), (0.5, 0.5), (0.4, 0.27)] + [tuple(x) for x in rng.uniform(0, 1, (12, 2))]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for a, b in ab:
            alpha, beta = upstream.ab_to_alphabeta(a, b)
            for n in (7, 17, 22):
                t, w = gauss_jacobi(n, beta - 1.0, alpha - 1.0)
                t2, w2 = bisection_gauss_jacobi(n, beta - 1.0, alpha - 1.0)
                assert np.abs(t - t2).max() < 2e-15, (a, b, n)
                assert np.abs(w - w2).max() < 1e-11, (a, b, n)      # (the bound of library vs LAPACK, test_upstream_grid.py)
                big = w > 1e-8
                assert np.abs(w2[big] / w[big] - 1).max() < 1e-10, (a, b, n)


def test_host_helpers_of_a_sample_batch():
    from starry_process_amd.calibrate import MAX_STREAMS, clamp_depth
    from starry_process_amd.engine import make_stars, sample_parameters, stars_for_samples

    sm = np.array([[20.0, 0.4, 0.27, 0.1, 10.0], [10.0, 0.0, 1.0, 0.05, 3.0]])
    p = sample_parameters(sm)
    for k, (r, a, b, c, n) in enumerate(sm):
        alpha, beta = upstream.ab_to_alphabeta(a, b)
        assert p[k, 0] == r * np.pi / 180 and p[k, 1] == alpha and p[k, 2] == beta and p[k, 3] == c and p[k, 4] == n
    for bad in ([95.0, 0.4, 0.27, 0.1, 10.0], [20.0, 1.1, 0.27, 0.1, 10.0], [20.0, 0.4, -0.1, 0.1, 10.0],
                [20.0, 0.4, 0.27, 0.1, -1.0], [20.0, 0.4, 0.27, np.nan, 1.0]):
        with pytest.raises(ValueError):
            sample_parameters([bad])
    with pytest.raises(ValueError):
        sample_parameters(np.zeros((2, 4)))
    from starry_process_amd.engine import samples_in_bounds

    rows = np.array([[20.0, 0.4, 0.27, 0.1, 10.0], [95.0, 0.4, 0.27, 0.1, 10.0], [20.0, 1.1, 0.27, 0.1, 10.0],
                     [20.0, 0.4, -0.1, 0.1, 10.0], [20.0, 0.4, 0.27, 0.1, -1.0], [20.0, 0.4, 0.27, np.nan, 1.0],
                     [0.0, 0.0, 1.0, -0.3, 0.0]])
    assert list(samples_in_bounds(rows)) == [True, False, False, False, False, False, True]
    for k, row in enumerate(rows):             # (the mask IS what sample_parameters raises for)
        try:
            sample_parameters([row])
            raised = False
        except ValueError:
            raised = True
        assert raised != bool(samples_in_bounds(rows)[k]), k
    stars = make_stars(3, period=[1.0, 2.0, 3.0], table=[0, 1, 0])
    rep = stars_for_samples(stars, 4, 2)
    assert rep.shape == (12,) and list(rep["table"]) == [0, 1, 0, 2, 3, 2, 4, 5, 4, 6, 7, 6]
    assert list(rep["period"]) == [1.0, 2.0, 3.0] * 4
    ab = np.random.RandomState(0).uniform(0, 1, (50, 2))
    v = upstream.log_jac_samples(ab[:, 0], ab[:, 1])
    w = np.array([upstream.log_jac(a, b) for a, b in ab])
    assert np.array_equal(np.isfinite(v), np.isfinite(w)) and np.allclose(v[np.isfinite(w)], w[np.isfinite(w)], rtol=1e-13, atol=1e-13)
    # depth + other streams never exceed MAX_STREAMS; one warning
    assert MAX_STREAMS == 4 and clamp_depth(3, 1) == 3 and clamp_depth(1) == 1 and clamp_depth(9, limit=6) == 6
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert clamp_depth(6, 1) == 3 and clamp_depth(9) == 4
    assert len(rec) <= 1
