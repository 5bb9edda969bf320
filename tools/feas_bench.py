"""Feasibility form on the device (fos_feas_*): iterations/s of DR on IndAffine(A, b) n IndBox(0, Inf), the achieved HBM rate of the
kernel that bounds it (the dense symmetric matrix-vector product P x, n^2 x 8 bytes per projection), and the same iteration in numpy
(one projection = two GEMVs + a Cholesky solve) timed on the host beside it.   python3 tools/feas_bench.py [n] [m] [iters]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge                                    # noqa: E402
from feasibility_cases import affine_box_instance              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
pkg = ge.load_package()
A, b = affine_box_instance(seed=7, m=m, n=n)
t0 = time.time()
dev = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n))
t_setup = time.time() - t0
dev.set_alg(pkg.DR())
dev.set_iterate(None)
dev.step(1, 20, 10 ** 9, 0.0)                                   # warm-up
t0 = time.perf_counter()
done, _, _, _ = dev.step(21, iters, 10 ** 9, 0.0)
dt = time.perf_counter() - t0
L = (n + 63) // 64 * 64
out = {"workload": "Feasibility: IndAffine(randn(%d,%d), b) n IndBox(0, Inf), DR" % (m, n), "iterations_per_s": done / dt,
       "ms_per_iteration": 1e3 * dt / done, "setup_s": t_setup, "newton_schulz_steps": dev.info()["ns_iters"][0],
       "bytes_per_iteration_dense_matvec": 8.0 * n * L,
       "whole_iteration_GBps_on_those_bytes": 8.0 * n * L / (dt / done) / 1e9, "peak_GBps": 8000.0}
# the same DR iteration in numpy on the host (Cholesky factor of A A' once, then two GEMVs + two triangular solves per projection)
import scipy.linalg
chol = scipy.linalg.cho_factor(A @ A.T)
x = np.zeros(n)
k = max(3, min(50, iters // 8))
t0 = time.perf_counter()
for i in range(k):
    p1 = x - A.T @ scipy.linalg.cho_solve(chol, A @ x - b)        # prox of IndAffine
    t1 = 2.0 * p1 - x                                             # DR = GAP(0.5, 2, 2)   solvers.jl:10, gap.jl:42-80
    t2 = 2.0 * np.maximum(t1, 0.0) - t1
    x = 0.5 * t2 + 0.5 * x
out["numpy_host_iterations_per_s"] = k / (time.perf_counter() - t0)
print(json.dumps(out))
