"""GPU: IndAffine(A, b) with a SPARSE A as a device-resident set of the Feasibility form (fos_feas_set_affine_sparse, csrc/affine_sparse.hip;
src/problemforms/Feasibility/Feasibility.jl:2-6 takes any ProximableFunction -- the reference's own test uses a dense IndAffine).  The projection is
unique, so the checks are against the oracle's dense Cholesky restatement (small sizes), a sparse LU of A A' (sizes beyond the dense projector's
n <= 46 000) and the dense device path; whole solves against the oracle's solve of the same problem."""
import numpy as np
import pytest
import scipy.sparse as sp

from feasibility_cases import ALGS

pytestmark = pytest.mark.gpu


def sparse_instance(seed, m, n, per_row, row_scale_decades=0.0):
    """A with `per_row` entries per row at random columns (+ one on a diagonal band: full row rank), b = A xs with xs >= 0 on the boundary"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(m), per_row)
    cols = rng.integers(0, n, m * per_row)
    vals = rng.standard_normal(m * per_row)
    band = (np.arange(m) * (n // m)) % n
    A = sp.csr_matrix((np.r_[vals, 2.0 + rng.random(m)], (np.r_[rows, np.arange(m)], np.r_[cols, band])), shape=(m, n))
    if row_scale_decades:
        A = sp.diags(10.0 ** rng.uniform(-row_scale_decades, row_scale_decades, m)) @ A
    xs = np.maximum(rng.standard_normal(n), 0.0)
    return sp.csc_matrix(A), A @ xs


@pytest.mark.parametrize("m,n,per_row,decades", [(300, 1000, 6, 0.0), (900, 1000, 3, 0.0), (1500, 6000, 12, 3.0), (40, 5000, 300, 0.0), (1, 50, 7, 0.0)])
def test_projection_matches_the_oracle_and_the_dense_device_path(pkg, oracle, m, n, per_row, decades):
    orc = oracle
    A, b = sparse_instance(m + n, m, n, per_row, decades)
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n))
    dd = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b, sparse=False), pkg.IndBox(0.0, np.inf), n))
    ref = orc.IndAffine(A.toarray(), b)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n)
    y = np.empty(n)
    its = []
    for rep in range(4):                                          # rep > 0: warm-started multipliers, slowly moving input
        ref.prox(y, x)
        yd = d.prox(1, x)
        st = d.affine_stats(1)
        its.append(st["last_cg_iterations"])
        scale = max(1.0, np.abs(x).max())
        assert np.abs(yd - y).max() <= 1e-12 * scale, (rep, st)
        assert np.abs(dd.prox(1, x) - yd).max() <= 1e-10 * scale      # (the dense projector is the less accurate of the two)
        rowscale = np.sqrt(np.asarray(A.multiply(A).sum(axis=1)).ravel())
        assert np.abs((A @ yd - b) / rowscale).max() <= 1e-13 * scale * np.sqrt(per_row + 1)
        assert st["last_residual"] <= 16 * 2.3e-16 * st["last_rounding_level"]
        x = x + 1e-3 * rng.standard_normal(n)
    assert its[0] > 0 and max(its[1:]) <= its[0]                  # the warm start pays
    assert np.array_equal(d.prox(1, x), d.prox(1, x))             # (same input, converged multipliers: the same bits)
    xl = 1e6 * x
    ref.prox(y, xl)
    assert np.abs(d.prox(1, xl) - y).max() <= 1e-12 * np.abs(xl).max()


def test_projection_beyond_the_dense_size(pkg):
    """n = 120 000 (the dense projector stops at 46 000).  A = 2000 diagonal blocks of 20 x 60 behind a row and a column permutation: the
    projection is block-wise, so the reference is 2000 small dense solves (a sparse LU of a random A A' of this size fills in for minutes)."""
    nb, bm, bn = 2000, 20, 60
    m, n = nb * bm, nb * bn
    rng = np.random.default_rng(7)
    blocks = rng.standard_normal((nb, bm, bn)) * (rng.random((nb, bm, bn)) < 0.3) + np.eye(bm, bn)[None]
    cperm, rperm = rng.permutation(n), rng.permutation(m)
    B = sp.block_diag([sp.csr_matrix(blk) for blk in blocks], format="csr")
    A = sp.csc_matrix(B[rperm][:, cperm])
    xs = np.maximum(rng.standard_normal(n), 0.0)
    b = A @ xs

    def reference(x):
        xb = np.empty(n); xb[cperm] = x                               # x in the blocks' column order
        bb = np.empty(m); bb[rperm] = b
        X, Bb = xb.reshape(nb, bn), bb.reshape(nb, bm)
        R = np.einsum("kij,kj->ki", blocks, X) - Bb
        G = np.einsum("kij,klj->kil", blocks, blocks)
        W = np.linalg.solve(G, R[..., None])[..., 0]
        Y = X - np.einsum("kij,ki->kj", blocks, W)
        return Y.reshape(n)[cperm]

    d = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n))
    x = rng.standard_normal(n)
    for rep in range(3):
        yd = d.prox(1, x)
        assert np.abs(yd - reference(x)).max() <= 1e-12 * max(1.0, np.abs(x).max()), d.affine_stats(1)
        x = x + 1e-2 * rng.standard_normal(n)
    with pytest.raises(pkg.lib.FosError):                          # a dense A of that width has no device form ...
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(A[:3], b[:3], sparse=False), pkg.IndBox(0.0, 1.0), n))


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA"])
def test_whole_solves_match_the_oracle(pkg, oracle, algname):
    orc = oracle
    m, n = 120, 400
    A, b = sparse_instance(3, m, n, 8)
    hp = pkg.Feasibility(pkg.IndAffine(A, b), pkg.IndBox(0.0, np.inf), n)
    op = orc.Feasibility(orc.IndAffine(A.toarray(), b), orc.IndBox(0.0, np.inf), n)
    kw = dict(eps=1e-9, max_iters=3000, verbose=0)
    sol, model = pkg.solve_feasibility(hp, ALGS[algname](pkg, **kw), checki=10)
    osol, _ = orc.feasibility_solve(op, ALGS[algname](orc, **kw), checki=10)
    assert sol.status == osol.status
    assert abs(sol.iterations - osol.iterations) <= 10
    assert np.abs(sol.x - osol.x).max() <= 1e-7
    if sol.status == "Optimal":
        assert sol.x.min() > -1e-8 and np.abs(A @ sol.x - b).max() < 1e-7


def test_error_paths(pkg):
    n = 30
    A, b = sparse_instance(5, 10, n, 4)
    A = sp.lil_matrix(A)
    A[7, :] = A[2, :]                                              # rank deficient with an inconsistent right-hand side: no projection exists
    b2 = b.copy(); b2[7] = b[2] + 1.0
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(sp.csc_matrix(A), b2), pkg.IndBox(0.0, 1.0), n))
    with pytest.raises(pkg.lib.FosError, match="rounding level"):
        d.prox(1, np.ones(n))
    Z = sp.lil_matrix(A); Z[4, :] = 0.0
    with pytest.raises(pkg.lib.FosError, match="zero"):
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndAffine(sp.csc_matrix(Z), b), pkg.IndBox(0.0, 1.0), n))
    with pytest.raises(ValueError):
        pkg.IndAffine(sp.csc_matrix(A), b[:-1])
    with pytest.raises(pkg.lib.FosError):
        pkg.HipFeasibility(pkg.Feasibility(pkg.IndBox(0.0, 1.0), pkg.IndBox(0.0, 1.0), n)).affine_stats(1)
