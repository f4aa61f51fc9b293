"""
TEST / BENCH INFRASTRUCTURE -- worker process of bench.py's cpu_baseline leg.

    python oracle/cpu_worker.py <moments.npz> <first_star> <last_star> <K> <ydeg> <udeg>

Evaluates OracleProcess.log_likelihood (the CPU restatement of the reference,
SciPy/LAPACK potrf + trtrs as in reference math.py:75-100) for the synthetic stars
[first, last) on ONE BLAS thread and prints one JSON line
{"stars": [...], "values": [...], "seconds": compute_time}.  Never touches the GPU.
"""
import json
import os
import sys
import time

for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[var] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import sp_oracle as orc  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402


def main():
    mom = np.load(sys.argv[1])
    first, last, K, ydeg, udeg = (int(a) for a in sys.argv[2:7])
    op = orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=ydeg, udeg=udeg)
    st = synthetic_star(0, K)
    op.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"])  # warm the constants
    t0 = time.perf_counter()
    vals = []
    for s in range(first, last):
        st = synthetic_star(s, K)
        vals.append(float(op.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"])))
    print(json.dumps({"stars": list(range(first, last)), "values": vals,
                      "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main()
