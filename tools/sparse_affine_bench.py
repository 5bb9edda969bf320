"""Feasibility form with a SPARSE IndAffine on the device (csrc/affine_sparse.hip): time and CG iterations of one projection, cold and in the
steady state of a DR solve, by problem size.  `python tools/sparse_affine_bench.py` -> one JSON object."""
import sys, time, json; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import __graft_entry__ as ge
from test_gpu_sparse_affine import sparse_instance
pkg = ge.load_package()
out = {}
for (m, n, per_row) in [(2000, 8000, 8), (40000, 120000, 5), (200000, 1000000, 10), (500000, 1000000, 10)]:
    A, b = sparse_instance(1, m, n, per_row)
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    d.prox(1, x)                                                   # (first call: code objects)
    d.set_iterate(None)
    t = time.perf_counter(); d.prox(1, x); cold_ms = 1e3 * (time.perf_counter() - t)
    cold = d.affine_stats(1)
    d.set_alg(pkg.DR()); d.set_iterate(x)
    d.step(1, 50, 10 ** 9, 0.0)
    s0 = d.affine_stats(1)
    t = time.perf_counter(); d.step(51, 100, 10 ** 9, 0.0); d.get_iterate(); dt = time.perf_counter() - t
    s1 = d.affine_stats(1)
    nnz = A.nnz
    its = (s1["cg_iterations"] - s0["cg_iterations"]) / 100.0
    out["m=%d n=%d nnz=%d" % (m, n, nnz)] = {
        "cold_projection_ms": round(cold_ms, 3), "cold_cg_iterations": cold["last_cg_iterations"], "cold_restarts": cold["last_restarts"],
        "dr_iteration_ms": round(1e3 * dt / 100, 4), "cg_iterations_per_projection": its, "lanes_per_row(A, A')": s1["lanes_per_row"],
        "last_residual": s1["last_residual"], "rounding_level_x_16eps": 16 * 2.22e-16 * s1["last_rounding_level"],
        "bytes_per_cg_iteration_MB": round((2 * nnz * 12 + 8 * (5 * m + 2 * n)) / 1e6, 2)}
    print(m, n, out["m=%d n=%d nnz=%d" % (m, n, nnz)], flush=True)
print(json.dumps({"what": "Feasibility(IndAffine(A sparse, b), IndBox(0, Inf), n) under DR on one MI355X; a projection = warm-started CG on the row-scaled "
                          "normal equations, three launches per iteration, verified by the recomputed residual (two more sweeps + a host look)",
                  "per_size": out}, indent=1))
