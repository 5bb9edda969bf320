"""Time of ONE batched PSD projection (cones.jl:11,89-94 -> IndPSD(scaling=true)) per matrix order and batch size.
`python tools/psd_orders.py [k ...]` -- for every order k and batch of `nc` cones (each cone = TWO matrices per projection: the primal copy s and the
dual copy y of the HSDE iterate) a handle with nc PSD(k) cones is built and prox!(., S2, .) is timed by HIP events on the solver's stream (profiling
class PSD): cold (first call on a fresh handle) and warm (the next point of a slowly drifting sequence, as in the solver's steady state; orders <= 64
keep their eigenvector bases).  Beside it: the flops of a projection done by matrix products (the refinement kernel's count at three iterations,
(3 x 3 + 3) products of 2 k^3 flops) on ONE CU's fp64 matrix cores -- the floor the review asked the Jacobi kernel to be held against."""
import sys; sys.path.insert(0, '.')
import json
import numpy as np
import scipy.sparse as sp
import __graft_entry__ as ge
pkg = ge.load_package()
ks = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 96, 128, 200, 256]
FP64_PEAK_PER_CU = 78.6e12 / 256
out = {}
rng = np.random.default_rng(0)
for k in ks:
    d_k = k * (k + 1) // 2
    for nc in (1, 64):
        m, n = nc * d_k, nc
        A = sp.csc_matrix((np.ones(nc), (np.arange(nc) * d_k, np.arange(nc))), shape=(m, n))      # (the operator does not matter here)
        dev = pkg.HipHSDE(A, np.zeros(m), np.zeros(n), [("SDP", d_k)] * nc, [("Free", n)])
        z0 = rng.standard_normal(dev.N)
        drift = rng.standard_normal(dev.N)
        dev.profile(1); dev.profile_read_classes()
        dev.prox_cones(z0)
        _, cold_ms = dev.profile_read_classes()["psd"]
        for q in range(1, 4):                                     # two untimed warm projections along the drift, then the timed ones
            dev.prox_cones(z0 + 1e-3 * q * drift)
        dev.profile_read_classes()
        reps = 5
        for q in range(4, 4 + reps):
            dev.prox_cones(z0 + 1e-3 * q * drift)
        nl, warm_ms = dev.profile_read_classes()["psd"]
        # parity spot check against LAPACK on the s copy of the first cone (svec, off-diagonals x sqrt 2)
        y = dev.prox_cones(z0)
        l = dev.l
        sv = z0[l + n:l + n + d_k]
        M = np.zeros((k, k)); iu = np.tril_indices(k); order = np.lexsort((iu[0], iu[1]))
        M[iu[0][order], iu[1][order]] = sv; M = M + M.T - np.diag(np.diag(M))
        off = ~np.eye(k, dtype=bool); M[off] /= np.sqrt(2.0)
        w, V = np.linalg.eigh(M); P = (V * np.maximum(w, 0)) @ V.T
        Pg = np.zeros((k, k)); Pg[iu[0][order], iu[1][order]] = y[l + n:l + n + d_k]; Pg = Pg + Pg.T - np.diag(np.diag(Pg)); Pg[off] /= np.sqrt(2.0)
        err = float(np.linalg.norm(Pg - P) / max(1e-300, np.linalg.norm(M)))
        floor_us = 1e6 * (3 * 3 + 3) * 2.0 * k ** 3 / FP64_PEAK_PER_CU * max(1.0, 2 * nc / 256.0)
        out["k=%d, %d cone%s (%d matrices)" % (k, nc, "" if nc == 1 else "s", 2 * nc)] = {
            "cold_us": round(1e3 * cold_ms, 1), "warm_us": round(1e3 * warm_ms / max(1, nl), 1),
            "matrix_product_floor_us": round(floor_us, 1), "warm_over_floor": round(1e3 * warm_ms / max(1, nl) / floor_us, 1),
            "rel_err_vs_lapack": err}
        dev.close()
        print(k, nc, out["k=%d, %d cone%s (%d matrices)" % (k, nc, "" if nc == 1 else "s", 2 * nc)], flush=True)
print(json.dumps({"what": "one prox!(., S2, .) over nc PSD(k) cones = 2 nc matrices; floor = 12 products of 2 k^3 flops on one CU's fp64 matrix cores "
                          "(78.6 TFLOP/s / 256), times the rounds of workgroups when there are more matrices than CUs",
                  "per_case": out}, indent=1))
