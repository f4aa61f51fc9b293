"""
TEST / BENCH INFRASTRUCTURE -- worker process of bench.py's cpu_baseline leg.

    python oracle/cpu_worker.py <moments.npz> <first_star> <last_star> <K> <ydeg> <udeg> [numpy|c]

Evaluates the log-likelihood of the synthetic stars [first, last) on ONE thread and prints one
JSON line {"stars": [...], "values": [...], "seconds": compute_time}.  Never touches the GPU.
  numpy (default): OracleProcess.log_likelihood, the NumPy/SciPy restatement of the reference
                   (LAPACK potrf + trtrs as in reference math.py:75-100);
  c:               oracle/cpu_pipeline.c, the same per-star pipeline in C (kernel table from the
                   NumPy oracle, LAPACK from the same SciPy), SURVEY.md 8d(i).
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
    engine = sys.argv[7] if len(sys.argv) > 7 else "numpy"
    op = orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=ydeg, udeg=udeg)
    if engine == "c":
        from oracle import cpu_pipeline as cp

        sts = [synthetic_star(s, K) for s in range(first, last)]
        args = ([s["t"] for s in sts], [s["flux"] for s in sts], [s["p"] for s in sts],
                [s["data_cov"] for s in sts])
        cp.lnlike(op, *[a[:1] for a in args], nthreads=1)    # warm (library load, first touch)
        t0 = time.perf_counter()
        vals, _, _ = cp.lnlike(op, *args, nthreads=1)         # (table build included, once per call)
        print(json.dumps({"stars": list(range(first, last)), "values": [float(v) for v in vals],
                          "seconds": time.perf_counter() - t0}))
        return
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
