"""
Synthetic conic problems for the BASELINE.json configs.

Every generator returns a `ConicProblem` in the MathProgBase conic form the reference's
`loadproblem!` receives (src/FOSSolverInterface.jl:27-64):

        minimize  c'x   subject to   b - A x in K1,   x in K2

with cones given as (name, length) pairs in order (contiguous, gap free -- the layout
`ConeProduct` asserts, src/cones.jl:66-72).  Data: numpy `default_rng(seed)` (PCG64), fp64.

Problems C2..C5 are built from a complementary primal-dual pair so that the optimum is known:
for every cone a random point z is split by Moreau's decomposition into s = P_K(z) and
y = P_K*(-z) (s in K, y in K*, s'y = 0); then b = A x0 + s0 and c = r0 - A'y0.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


@dataclass
class ConicProblem:
    name: str
    A: sp.csc_matrix
    b: np.ndarray
    c: np.ndarray
    K1: list            # [(cone name, length), ...] for the rows   (constr_cones)
    K2: list            # [(cone name, length), ...] for the columns (var_cones)
    x0: np.ndarray | None = None      # a known optimal primal point (None if unknown)
    y0: np.ndarray | None = None
    s0: np.ndarray | None = None
    meta: dict = field(default_factory=dict)

    @property
    def m(self):
        return self.A.shape[0]

    @property
    def n(self):
        return self.A.shape[1]

    @property
    def nnz(self):
        return self.A.nnz


# ---------------------------------------------------------------------------- cone helpers


def _svec(M):
    """Lower triangle column-major, off-diagonals times sqrt(2)."""
    k = M.shape[0]
    out = np.empty(k * (k + 1) // 2)
    idx = 0
    r2 = math.sqrt(2.0)
    for j in range(k):
        out[idx] = M[j, j]
        out[idx + 1:idx + k - j] = M[j + 1:, j] * r2
        idx += k - j
    return out


def _psd_order(length):
    k = int(round(math.sqrt(0.25 + 2.0 * length) - 0.5))
    assert k * (k + 1) // 2 == length
    return k


def _moreau_pair(rng, name, length):
    """(s, y) with s in K, y in K*, s'y = 0, neither trivially zero where the cone allows it."""
    if name == "Zero":
        return np.zeros(length), rng.standard_normal(length)
    if name == "Free":
        return rng.standard_normal(length), np.zeros(length)
    if name == "NonNeg":
        z = rng.standard_normal(length)
        return np.maximum(z, 0.0), np.maximum(-z, 0.0)
    if name == "NonPos":
        z = rng.standard_normal(length)
        return np.minimum(z, 0.0), np.minimum(-z, 0.0)
    if name == "SOC":
        z = rng.standard_normal(length)
        z[0] *= 0.5                        # mostly the "neither inside nor in the polar" case
        t, v = z[0], z[1:]
        nv = np.linalg.norm(v)

        def proj(t, v, nv):
            if t <= -nv:
                return np.zeros(length)
            if t >= nv:
                return np.concatenate([[t], v])
            r = 0.5 * (1 + t / nv)
            return np.concatenate([[r * nv], r * v])
        return proj(t, v, nv), proj(-t, -v, nv)
    if name == "SDP":
        k = _psd_order(length)
        G = rng.standard_normal((k, k))
        Z = (G + G.T) / 2
        lam, V = np.linalg.eigh(Z)
        S = (V * np.maximum(lam, 0)) @ V.T
        Y = (V * np.maximum(-lam, 0)) @ V.T
        return _svec(S), _svec(Y)
    raise ValueError(name)


def _pairs_for(rng, cones):
    ss, ys = [], []
    for name, length in cones:
        s, y = _moreau_pair(rng, name, length)
        ss.append(s)
        ys.append(y)
    return (np.concatenate(ss) if ss else np.zeros(0)), (np.concatenate(ys) if ys else np.zeros(0))


TARGET_NORM = 10.0     # ||b|| and ||c|| of the generated problems (see normalize_data)


def normalize_data(x0, s0, y0, r0, A, target=TARGET_NORM):
    """The reference does not equilibrate its data, and the HSDE iteration started from tau = kappa = 1 spends
    hundreds of iterations at tau = 0 when ||b||, ||c|| are in the hundreds.  Cones are positively homogeneous, so
    scaling the primal pair (x0, s0) and the dual pair (y0, r0) by positive constants gives an equivalent problem:
    pick the constants that make ||b|| = ||c|| = target."""
    b = A @ x0 + s0
    c = r0 - A.T @ y0
    nb, nc = np.linalg.norm(b), np.linalg.norm(c)
    sp_, sd_ = (target / nb if nb > 0 else 1.0), (target / nc if nc > 0 else 1.0)
    return x0 * sp_, s0 * sp_, y0 * sd_, r0 * sd_, b * sp_, c * sd_


def from_complementary_pair(name, A, K1, K2, rng, meta=None, normalize=True):
    """b = A x0 + s0, c = r0 - A'y0 with (s0,y0) and (x0,r0) complementary cone pairs."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    s0, y0 = _pairs_for(rng, K1)
    x0, r0 = _pairs_for(rng, K2)
    if normalize:
        x0, s0, y0, r0, b, c = normalize_data(x0, s0, y0, r0, A)
    else:
        b = A @ x0 + s0
        c = r0 - A.T @ y0
    return ConicProblem(name, A, b, c, list(K1), list(K2), x0=x0, y0=y0, s0=s0, meta=meta or {})


# ---------------------------------------------------------------------------- configs


def c1_readme_nnls(seed=0, m=40, n=50, data=None):
    """C1: README least squares  min ||Ax-b||^2 s.t. x >= 0  (README.md:21-26, test/testDRandGAPA.jl:4-8),
    emitted directly in conic form: vars (x, t, w), min w,
    (t, Ax-b) in SOC(m+1), (w+1, w-1, 2t) in SOC(3), x in NonNeg(n); K2 = Free.
    `data=(A, b)` takes the dense data from the caller (the reference's own draws, tests/golden/reference_test_inputs.npz)."""
    if data is not None:
        Ad = np.asarray(data[0], dtype=np.float64)
        bd = np.asarray(data[1], dtype=np.float64).reshape(-1)
        m, n = Ad.shape
    else:
        rng = np.random.default_rng(seed)
        Ad = rng.standard_normal((m, n))
        bd = rng.standard_normal(m)
    nv = n + 2
    it, iw = n, n + 1
    rows, cols, vals = [], [], []

    def add(r, c_, v):
        rows.append(r)
        cols.append(c_)
        vals.append(v)
    bb = np.zeros(m + 1 + 3 + n)
    # SOC(m+1): [t; A x - b] = b1 - A1 xc
    add(0, it, -1.0)
    for i in range(m):
        for j in range(n):
            add(1 + i, j, -Ad[i, j])
        bb[1 + i] = -bd[i]
    # SOC(3): [w+1; w-1; 2t]
    r0 = m + 1
    add(r0, iw, -1.0)
    bb[r0] = 1.0
    add(r0 + 1, iw, -1.0)
    bb[r0 + 1] = -1.0
    add(r0 + 2, it, -2.0)
    # NonNeg(n): x
    r1 = r0 + 3
    for j in range(n):
        add(r1 + j, j, -1.0)
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m + 4 + n, nv))
    A.sort_indices()
    c = np.zeros(nv)
    c[iw] = 1.0
    K1 = [("SOC", m + 1), ("SOC", 3), ("NonNeg", n)]
    K2 = [("Free", nv)]
    return ConicProblem("C1-readme-nnls", A, bb, c, K1, K2, meta=dict(Ad=Ad, bd=bd, n=n, m=m))


def c2_lp(seed=1, m=5000, n=10000, scale=100.0):
    """C2: standard form LP, dense A stored sparse (loadproblem! sparsifies, FOSSolverInterface.jl:27-29),
    K1 = Zero(m), K2 = NonNeg(n); x0 has m non-zeros with a complementary r0."""
    rng = np.random.default_rng(seed)
    Ad = rng.standard_normal((m, n)) / scale
    A = sp.csc_matrix(Ad)
    del Ad
    A.sort_indices()
    K1 = [("Zero", m)]
    K2 = [("NonNeg", n)]
    y0 = rng.standard_normal(m)
    s0 = np.zeros(m)
    x0 = np.zeros(n)
    r0 = np.zeros(n)
    basis = rng.permutation(n)[:min(m, n)]
    mask = np.zeros(n, dtype=bool)
    mask[basis] = True
    x0[mask] = rng.uniform(0.5, 1.5, size=int(mask.sum()))
    r0[~mask] = rng.uniform(0.5, 1.5, size=int((~mask).sum()))
    x0, s0, y0, r0, b, c = normalize_data(x0, s0, y0, r0, A)
    return ConicProblem("C2-lp-dense-%dx%d" % (m, n), A, b, c, K1, K2, x0=x0, y0=y0, s0=s0,
                        meta=dict(scale=scale))


def c3_socp(seed=2, n=20000, ncones=1000, conedim=50, density=1e-3):
    """C3: sparse SOCP, n free variables, K1 = ncones x SOC(conedim), A = sprandn(m, n, density)."""
    rng = np.random.default_rng(seed)
    m = ncones * conedim
    A = sp.random(m, n, density=density, format="csc", random_state=rng,
                  data_rvs=rng.standard_normal)
    K1 = [("SOC", conedim)] * ncones
    K2 = [("Free", n)]
    return from_complementary_pair("C3-socp-%dx%d" % (m, n), A, K1, K2, rng,
                                   meta=dict(density=density))


def c4_block_sdp(seed=3, nblocks=512, k=64, p=32, scale=None, block_range=None):
    """C4: block-diagonal SDP.  Block j: p free variables, K1 block = PSD(k) (svec dim k(k+1)/2),
    A_j dense (k(k+1)/2) x p whose columns are svec'd random symmetric matrices, divided by `scale`
    (default k/2, which puts the singular values of A_j near 1.4).  `block_range=(lo,hi)` builds only
    those blocks (identical numbers to the same blocks of the full problem) -- used by cone sharding."""
    d = k * (k + 1) // 2
    if scale is None:
        scale = k / 2.0
    lo, hi = (0, nblocks) if block_range is None else block_range
    nb = hi - lo
    datas, x0s, s0s, y0s = [], [], [], []
    r2 = math.sqrt(2.0)
    tri = np.tril_indices(k)
    # column-major lower triangle order: sort (col, row)
    order = np.lexsort((tri[0], tri[1]))
    ti, tj = tri[0][order], tri[1][order]
    offd = ti != tj
    for j in range(lo, hi):
        rng = np.random.default_rng([seed, j])            # per-block stream: shard == slice of the whole
        G = rng.standard_normal((p, k, k))
        Ssym = (G + np.transpose(G, (0, 2, 1))) / 2
        cols = Ssym[:, ti, tj]                            # (p, d)
        cols[:, offd] *= r2
        datas.append((cols / scale).T.copy())             # (d, p)
        s, y = _moreau_pair(rng, "SDP", d)
        s0s.append(s)
        y0s.append(y)
        x0s.append(rng.standard_normal(p))
    A = sp.block_diag([sp.csc_matrix(D) for D in datas], format="csc")
    A.sort_indices()
    # data scaling (see normalize_data): per-block norms are ~ k/2 for b and ~ k/12 for c, so these constants put
    # ||b|| and ||c|| near TARGET_NORM for any number of blocks; a shard only needs the GLOBAL block count.
    sig_p = TARGET_NORM / (0.52 * k * math.sqrt(nblocks))
    sig_d = TARGET_NORM / (0.086 * k * math.sqrt(nblocks))
    x0 = np.concatenate(x0s) * sig_p
    s0 = np.concatenate(s0s) * sig_p
    y0 = np.concatenate(y0s) * sig_d
    b = A @ x0 + s0
    c = -(A.T @ y0)                                       # K2 = Free  ->  r0 = 0
    K1 = [("SDP", d)] * nb
    K2 = [("Free", p * nb)]
    return ConicProblem("C4-blocksdp-%dx%d" % (nblocks, k), A, b, c, K1, K2, x0=x0, y0=y0, s0=s0,
                        meta=dict(nblocks=nblocks, k=k, p=p, scale=scale, block_range=(lo, hi)))


def c5_mixed(seed=4, nblocks=8, nb_cols=62500, nonneg=31250, nsoc=250, socdim=50, npsd=9, k=64,
             density=3.2e-4, block_range=None):
    """C5: mixed-cone block problem.  Block j: nb_cols free variables,
    K1_j = NonNeg(nonneg) + nsoc x SOC(socdim) + npsd x PSD(k), A_j = sprandn(m_b, nb_cols, density)."""
    d = k * (k + 1) // 2
    mb = nonneg + nsoc * socdim + npsd * d
    lo, hi = (0, nblocks) if block_range is None else block_range
    blocks, K1, x0s, s0s, y0s = [], [], [], [], []
    for j in range(lo, hi):
        rng = np.random.default_rng([seed, j])
        Aj = sp.random(mb, nb_cols, density=density, format="csc", random_state=rng,
                       data_rvs=rng.standard_normal)
        blocks.append(Aj)
        K1j = [("NonNeg", nonneg)] + [("SOC", socdim)] * nsoc + [("SDP", d)] * npsd
        s, y = _pairs_for(rng, K1j)
        K1 += K1j
        s0s.append(s)
        y0s.append(y)
        x0s.append(rng.standard_normal(nb_cols))
    A = sp.block_diag(blocks, format="csc")
    A.sort_indices()
    # deterministic data scaling from global quantities only (see normalize_data / c4_block_sdp)
    est_b = math.sqrt(nblocks * mb * (1.0 + density * nb_cols))
    est_c = math.sqrt(nblocks * nb_cols * density * mb * 0.5)
    sig_p, sig_d = TARGET_NORM / est_b, TARGET_NORM / est_c
    x0 = np.concatenate(x0s) * sig_p
    s0 = np.concatenate(s0s) * sig_p
    y0 = np.concatenate(y0s) * sig_d
    b = A @ x0 + s0
    c = -(A.T @ y0)
    K2 = [("Free", nb_cols * (hi - lo))]
    return ConicProblem("C5-mixed-%dblocks" % nblocks, A, b, c, K1, K2, x0=x0, y0=y0, s0=s0,
                        meta=dict(nblocks=nblocks, block_range=(lo, hi), mb=mb, nb_cols=nb_cols))


def psd2x2_reference_problem():
    """test/testPSD.jl:3-4,22-25:  minimize ||vec(Y - ys)||  s.t.  Y PSD, ys the fixed 2x2 matrix.
    Conic form: vars (t, v = svec(Y));  (t, v - svec(ys)) in SOC(4),  v in SDP(3);  K2 = Free(4)."""
    ys = np.array([[-0.0064709, -0.22443], [-0.22443, -1.02411]])
    vs = _svec(ys)
    Ad = np.zeros((7, 4))
    Ad[0:4, 0:4] = -np.eye(4)          # SOC(4) rows:  [t; v - vs] = b - A x
    Ad[4:7, 1:4] = -np.eye(3)          # SDP(3) rows:  v = b - A x
    A = sp.csc_matrix(Ad)
    b = np.concatenate([[0.0], -vs, np.zeros(3)])
    c = np.array([1.0, 0, 0, 0])
    return ConicProblem("testPSD-2x2", A, b, c, [("SOC", 4), ("SDP", 3)], [("Free", 4)], meta=dict(ys=ys))


def small_lp(seed=11, m=30, n=60):
    return c2_lp(seed=seed, m=m, n=n, scale=math.sqrt(n))


def small_mixed(seed=12):
    """A small problem touching every supported cone kind on both the row and the column side."""
    rng = np.random.default_rng(seed)
    K1 = [("Zero", 5), ("NonNeg", 7), ("SOC", 6), ("SDP", 10), ("SOC", 3), ("NonPos", 4), ("SDP", 6), ("Free", 2)]
    K2 = [("Free", 8), ("NonNeg", 9), ("SOC", 5), ("Zero", 2), ("NonPos", 3)]
    m = sum(l for _, l in K1)
    n = sum(l for _, l in K2)
    A = sp.random(m, n, density=0.3, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    return from_complementary_pair("small-mixed", A, K1, K2, rng)


CONFIGS = {
    "C1": c1_readme_nnls,
    "C2": c2_lp,
    "C3": c3_socp,
    "C4": c4_block_sdp,
    "C5": c5_mixed,
}


def mid_mixed(seed=14):
    """l ~ 1e4 mixed-cone problem (NonNeg + SOC(50) + PSD(16), 2 blocks, ~20 non-zeros per row): large enough for the device
    paths of the full-size configurations (ELL blocks, batched SOC/PSD kernels, warm-started PSD), small enough for the numpy
    oracle to run whole solves in seconds -- the mid-size whole-solve parity case."""
    return c5_mixed(seed=seed, nblocks=2, nb_cols=2500, nonneg=1200, nsoc=12, socdim=50, npsd=2, k=16, density=8e-3)
