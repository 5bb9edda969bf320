#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (oracle/fos_oracle.py).  The Julia reference cannot be executed in
the build container, so these vectors pin the ORACLE's outputs (which tests/test_oracle_reference.py ties to the
reference's own tests); the GPU tests compare the HIP path against them without importing the oracle.
Run from the repo root:  python tests/golden/make_golden.py   (and `... make_golden.py mid` for the three whole solves of the
l ~ 1e4 problem, ~5 minutes)"""
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


def mid_size_solves():
    """The whole solves of the l ~ 1e4 mixed-cone problem (workloads.mid_mixed) that tests/test_gpu_certificates.py compares the device
    with: the numpy oracle needs 1-2 minutes for each, so its results are kept here instead of being recomputed on the GPU box at
    every run (the problem itself is regenerated from its seed, not stored).  tests/test_golden.py re-derives the stored end-point
    residuals with the oracle's formulas (cheap) to tie the file to the oracle."""
    prob = pkg.workloads.mid_mixed()
    mks = {"DR": lambda **o: orc.DR(**o), "GAPA": lambda **o: orc.GAPA(0.8, 0.5, **o), "FISTA": lambda **o: orc.FISTA(**o)}
    out = {}
    checked = {}
    inner = orc.HSDEStatus.checkstatus

    def keep(self, z, override=False):                 # the point of the LAST evaluated check (the one `last` belongs to)
        done = inner(self, z, override=override)
        if self.checked:
            checked["z"] = np.array(z, copy=True)
        return done
    orc.HSDEStatus.checkstatus = keep
    for name, mk in mks.items():
        opts = dict(eps={"DR": 1e-6, "GAPA": 1e-5, "FISTA": 1e-6}[name], verbose=0, max_iters=2500 if name != "FISTA" else 300, checki=100)
        mo = orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))
        sol = orc.solve(mo, mk(**opts), out=[])
        last = sol.status_obj.last
        out[name + "_x"] = sol.x
        out[name + "_zchecked"] = checked["z"]
        out[name + "_status"] = np.array([sol.status])
        out[name + "_iterations"] = sol.iterations
        out[name + "_obj"] = sol.obj_val
        out[name + "_pdg"] = np.array([last["p"], last["d"], last["g"]])
        out[name + "_opts"] = np.array([opts["eps"], opts["checki"], opts["max_iters"]])
        print("mid_mixed", name, sol.status, sol.iterations, out[name + "_pdg"])
    orc.HSDEStatus.checkstatus = inner
    np.savez_compressed(OUT / "mid_mixed_solves.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        mid_size_solves()
        raise SystemExit(0)
    operators("small_mixed", pkg.workloads.small_mixed(), 100)
    operators("c4_tiny", pkg.workloads.c4_block_sdp(nblocks=3, k=6, p=4), 101)
    operators("c1_nnls", pkg.workloads.c1_readme_nnls(seed=2), 102)
    solve("c1_nnls_dr", pkg.workloads.c1_readme_nnls(seed=2), orc.DR, eps=1e-8, checki=100, max_iters=10000, verbose=0)
    solve("psd2x2_dr", pkg.workloads.psd2x2_reference_problem(), orc.DR, eps=1e-8, checki=100, max_iters=10000, verbose=0)
    solve("small_mixed_dr", pkg.workloads.small_mixed(), orc.DR, eps=1e-6, checki=50, max_iters=3000, verbose=0)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
