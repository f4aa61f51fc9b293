"""
Parity at every size of bench.py's K sweep (`other_shapes.k_sweep`; the reference's own benchmark protocol sweeps the
number of cadences, joss/figures/speed.py:22-37): K = 64 ... 4096 at ydeg 15, the batch sizes the sweep runs (about
0.5 GB of systems), marginal branch through the planned and the unplanned call, conditional branch -- a few stars of
each batch against the CPU oracle (1e-8, BASELINE.json), every star's planned value against its unplanned one, and a
star's value independent of the batch it is evaluated in.
"""
import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def engine():
    from starry_process_amd.engine import Engine

    e = Engine(15, 2, 0)
    mom = golden("moments_L15")
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    return e


def sweep_stars(K):
    import bench

    return bench.k_sweep_stars(K)


@pytest.mark.parametrize("K", [64, 128, 256, 512, 1000, 2048, 4096])
def test_k_sweep_sizes_against_the_oracle(engine, K):
    from oracle.sp_oracle import OracleProcess
    from starry_process_amd.engine import make_stars

    e = engine
    S = min(sweep_stars(K), 512)          # (the sweep's batch, capped: the host builds every light curve)
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
    stars = make_stars(S, period=[s["p"] for s in sts], inc_deg=[s["i"] for s in sts], data_var=1e-6)
    s_d = e.stars_to_device(stars)
    rta1 = e.f64(e.rTA1L([0.0, 0.0]))
    tab, mv = e.kernel_table(rta1, 300)
    un, st_u = e.lnlike_ensemble(t_d, f_d, s_d, covpts=300, tab=tab, meanvar=mv, normalized=True)
    un = un.cpu().numpy().copy()
    plan = e.plan_data(t_d, f_d, s_d, covpts=300)
    pl, st_p = e.lnlike_ensemble_planned(plan, t_d, f_d, s_d, tab, mv)
    pl = pl.cpu().numpy().copy()
    cond, st_c = e.lnlike_ensemble(t_d, f_d, s_d, conditional=True, rta1=rta1, normalized=True)
    cond = cond.cpu().numpy().copy()
    assert not st_u.cpu().numpy().any() and not st_p.cpu().numpy().any() and not st_c.cpu().numpy().any()
    assert np.all(np.isfinite(un)) and np.all(np.isfinite(cond))
    assert np.max(np.abs(pl / un - 1)) < 1e-10
    mom = golden("moments_L15")
    om = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    oc = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, marginalize_over_inclination=False)
    for s in sorted({0, S // 2, S - 1}):
        ref = om.log_likelihood(sts[s]["t"], sts[s]["flux"], 1e-6, p=sts[s]["p"])
        assert abs(un[s] / ref - 1) < TOL and abs(pl[s] / ref - 1) < TOL, (K, s, un[s], pl[s], ref)
        if K <= 2048 or s == 0:
            refc = oc.log_likelihood(sts[s]["t"], sts[s]["flux"], 1e-6, i=sts[s]["i"], p=sts[s]["p"])
            assert abs(cond[s] / refc - 1) < TOL, (K, s, cond[s], refc)
    # a star's value does not depend on the batch around it
    sub = slice(S // 2, S // 2 + 1)
    one, _ = e.lnlike_ensemble(t_d[sub].contiguous(), f_d[sub].contiguous(), e.stars_to_device(stars[sub]), covpts=300,
                               tab=tab, meanvar=mv, normalized=True)
    assert float(one[0]) == un[S // 2]
