#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (oracle/fos_oracle.py).  The Julia reference cannot be executed in
the build container, so these vectors pin the ORACLE's outputs (which tests/test_oracle_reference.py ties to the
reference's own tests); the GPU tests compare the HIP path against them without importing the oracle.
Run from the repo root:  python tests/golden/make_golden.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import __graft_entry__ as ge          # noqa: E402
import fos_oracle as orc              # noqa: E402

pkg = ge.load_package()
OUT = Path(__file__).resolve().parent


def codes(cs):
    return [(orc.CONE_CODES[k], l) for k, l in cs]


def pack_problem(prob):
    A = prob.A.tocsc()
    return dict(m=prob.m, n=prob.n, indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int64), data=A.data,
                b=prob.b, c=prob.c, K1=np.array([[orc.CONE_CODES[k], l] for k, l in prob.K1], dtype=np.int64),
                K2=np.array([[orc.CONE_CODES[k], l] for k, l in prob.K2], dtype=np.int64))


def operators(name, prob, seed):
    rng = np.random.default_rng(seed)
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    l = prob.m + prob.n + 1
    Q = orc.HSDEMatrixQ(prob.A, prob.b, prob.c)
    M = orc.KKTMatrix(Q)
    S2 = orc.DualConeProduct(mo.K1, mo.K2)
    xl = rng.standard_normal(l)
    xN = rng.standard_normal(2 * l)
    q_out, qt_out, kkt_out, cone_out = np.empty(l), np.empty(l), np.empty(2 * l), np.empty(2 * l)
    Q.mul(q_out, xl)
    Q.mul_t(qt_out, xl)
    M.mul(kkt_out, xN)
    S2.prox(cone_out, xN)
    zc = xN.copy()
    zc[l - 1] = abs(zc[l - 1]) + 0.5
    res = orc.residuals(mo, zc)
    np.savez_compressed(OUT / ("%s_operators.npz" % name), **pack_problem(prob), xl=xl, xN=xN, q_out=q_out, qt_out=qt_out,
                        kkt_out=kkt_out, cone_out=cone_out, zc=zc,
                        res=np.array([res[k] for k in ("p", "d", "g", "ctx", "bty", "kappa", "tau", "nAxs", "nATy", "nb", "nc")]))


def solve(name, prob, mk, **opts):
    mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
    sol = orc.solve(mo, mk(**opts), out=[])
    hist = {k: np.array([v for _, v in mo.history[k]]) for k in ("p", "d", "g", "ctx", "bty", "tau", "kappa")}
    np.savez_compressed(OUT / ("%s_solve.npz" % name), **pack_problem(prob), x=sol.x, y=sol.y, s=sol.s,
                        status=np.array([sol.status]), iterations=sol.iterations, obj=sol.obj_val,
                        hist_iter=np.array([i for i, _ in mo.history["p"]]), **{"hist_" + k: v for k, v in hist.items()},
                        opts=np.array([opts["eps"], opts["checki"], opts["max_iters"]]))


if __name__ == "__main__":
    operators("small_mixed", pkg.workloads.small_mixed(), 100)
    operators("c4_tiny", pkg.workloads.c4_block_sdp(nblocks=3, k=6, p=4), 101)
    operators("c1_nnls", pkg.workloads.c1_readme_nnls(seed=2), 102)
    solve("c1_nnls_dr", pkg.workloads.c1_readme_nnls(seed=2), orc.DR, eps=1e-8, checki=100, max_iters=10000, verbose=0)
    solve("psd2x2_dr", pkg.workloads.psd2x2_reference_problem(), orc.DR, eps=1e-8, checki=100, max_iters=10000, verbose=0)
    solve("small_mixed_dr", pkg.workloads.small_mixed(), orc.DR, eps=1e-6, checki=50, max_iters=3000, verbose=0)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
