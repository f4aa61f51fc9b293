"""
oracle/cpu_pipeline.c (the C restatement of the per-star pipeline behind bench.py's cpu_baseline)
against the golden values of the executed reference and against the NumPy oracle.  CPU only.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import cpu_pipeline as cp
from oracle import sp_oracle as orc
from starry_process_amd.synthetic import synthetic_star


@pytest.fixture(scope="module")
def op():
    mom = golden("moments_L15")
    return orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, udeg=2)


def test_matches_reference_golden_K200(op):
    g = golden("lnlike")
    idx = [int(s) for s in g["L15_K200_stars"]]
    sts = [synthetic_star(s, 200) for s in idx]
    v, secs, used = cp.lnlike(op, [s["t"] for s in sts], [s["flux"] for s in sts], [s["p"] for s in sts],
                              [s["data_cov"] for s in sts], nthreads=1)
    assert used == 1 and secs > 0
    assert np.max(np.abs(v / g["L15_K200"] - 1)) < 1e-9


def test_matches_reference_golden_K1000(op):
    g = golden("lnlike")
    idx = [int(s) for s in g["cfg2_L15_K1000_stars"]][:2]
    sts = [synthetic_star(s, 1000) for s in idx]
    v, _, _ = cp.lnlike(op, [s["t"] for s in sts], [s["flux"] for s in sts], [s["p"] for s in sts], 1e-6,
                        nthreads=1)
    assert np.max(np.abs(v / g["cfg2_L15_K1000"][:2] - 1)) < 1e-9


def test_matches_numpy_oracle_and_failure(op):
    sts = [synthetic_star(s, 120) for s in range(30, 34)]
    v, _, _ = cp.lnlike(op, [s["t"] for s in sts], [s["flux"] for s in sts], [s["p"] for s in sts],
                        [1e-6, 1e-6, -1.0, 1e-6], nthreads=2)
    ref = [op.log_likelihood(s["t"], s["flux"], 1e-6, p=s["p"]) for s in sts]
    ok = [0, 1, 3]
    assert np.max(np.abs(v[ok] / np.array(ref)[ok] - 1)) < 1e-10
    assert v[2] == -np.inf      # not positive definite -> -inf (math.py:82-91, sp.py:1186-1188)
