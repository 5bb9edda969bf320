"""
Cone sharding of block-separable problems across GPUs.

The reference has no parallelism at all (`#TODO Paralell implementation`, src/cones.jl:90,107); what makes the
hot path shardable is that S2 is a product of independent cones (src/cones.jl:89-94) and, when A is block
diagonal conformally with those cones, S1's operator couples shards only through
  * the tau row/column of Q  (-c'x - b'y,  HSDEAffine.jl:57)  and
  * the inner products of CG (conjugategradients.jl:35,39,42,46), GAPA's angle estimate (gapa.jl:36-47) and the
    residual norms of checkstatus (HSDEStatus.jl:34-38).
So rank g owns a contiguous run of whole K1 cones, the matching rows of A, b, y, s and the columns those rows
touch (x, r, c); tau and kappa are replicated; every reduction point all-reduces a handful of doubles (RCCL on the
GPU, gloo in the CPU tests) and adds the replicated tau/kappa contribution once, after the all-reduce.

`shard_problem` cuts a global ConicProblem; generators with a `block_range` argument (workloads.c4_block_sdp,
c5_mixed) can build a shard directly without ever materialising the global problem.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from .workloads import ConicProblem

_ELEMENTWISE = ("Free", "Zero", "NonNeg", "NonPos")


@dataclass
class Shard:
    problem: ConicProblem          # the local problem handed to fos_create on this rank
    rows: tuple                    # (r0, r1) global row range
    cols: tuple                    # (c0, c1) global column range
    m_global: int
    n_global: int


def _cone_bounds(cones):
    b = [0]
    for _, ln in cones:
        b.append(b[-1] + int(ln))
    return np.asarray(b, dtype=np.int64)


def balanced_cone_split(weights, nranks):
    """Contiguous split of len(weights) items into nranks runs with near-equal total weight.
    Returns nranks+1 item indices."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for g in range(1, nranks):
        target = cum[-1] * g / nranks
        j = int(np.searchsorted(cum, target))
        if j > 0 and (j >= len(cum) or target - cum[j - 1] <= cum[j] - target):
            j -= 1
        j = min(max(j, cuts[-1] + 1), len(w) - (nranks - g))
        cuts.append(j)
    cuts.append(len(w))
    return cuts


def _split_cones_at(cones, cut_points):
    """Cut a cone list at absolute index positions; only elementwise cones may be cut inside."""
    out = [[] for _ in range(len(cut_points) - 1)]
    pos = 0
    g = 0
    for name, ln in cones:
        ln = int(ln)
        start, end = pos, pos + ln
        while start < end:
            while g + 1 < len(cut_points) - 1 and cut_points[g + 1] <= start:
                g += 1
            seg_end = min(end, cut_points[g + 1])
            if (seg_end - start) != ln and name not in _ELEMENTWISE:
                raise ValueError("column split at %d falls inside a %s cone: the problem is not block separable"
                                 % (seg_end, name))
            if seg_end > start:
                out[g].append((name, seg_end - start))
            start = seg_end
        pos = end
    return out


def _separable_cone_cuts(A_csr, kb, K2):
    """K1-cone indices i (0 < i < #cones) at which the problem separates: every row before cone i lies left of every row from cone i on, and the
    column where the right part starts does not fall inside a second-order / PSD / exponential cone of K2."""
    m = A_csr.shape[0]
    nnz_per_row = np.diff(A_csr.indptr)
    has = nnz_per_row > 0
    rmax = np.full(m, -1, dtype=np.int64)
    rmin = np.full(m, np.iinfo(np.int64).max, dtype=np.int64)
    if A_csr.nnz:
        starts = A_csr.indptr[:-1][has]
        rmin[has] = np.minimum.reduceat(A_csr.indices, starts)
        rmax[has] = np.maximum.reduceat(A_csr.indices, starts)
    pmax = np.maximum.accumulate(rmax) if m else rmax                      # largest column of rows 0..r
    smin = np.minimum.accumulate(rmin[::-1])[::-1] if m else rmin          # smallest column of rows r..
    k2b = set(_cone_bounds(K2).tolist())
    inner_ok = []                                                          # columns at which K2 may be cut: cone boundaries, or inside elementwise cones
    pos = 0
    for name, ln in K2:
        inner_ok.append((pos, pos + int(ln), name in _ELEMENTWISE))
        pos += int(ln)

    def col_cut_ok(c):
        if c in k2b:
            return True
        return any(lo < c < hi and ew for lo, hi, ew in inner_ok)
    out = set()
    for i in range(1, len(kb) - 1):
        r = int(kb[i])
        if r <= 0 or r >= m:
            continue
        left = int(pmax[r - 1])
        right = int(smin[r]) if smin[r] != np.iinfo(np.int64).max else A_csr.shape[1]
        if left < right and col_cut_ok(right):
            out.add(i)
    return out


def plan(problem: ConicProblem, nranks: int):
    """Row / column cut points for an nranks-way cone sharding; raises ValueError if A is not block diagonal
    conformally with the K1 cones."""
    A = problem.A.tocsr()
    m, n = A.shape
    kb = _cone_bounds(problem.K1)
    if len(problem.K1) < nranks:
        raise ValueError("only %d K1 cones for %d ranks" % (len(problem.K1), nranks))
    nnz_per_row = np.diff(A.indptr)
    w = [float(nnz_per_row[kb[i]:kb[i + 1]].sum()) + 4.0 * (kb[i + 1] - kb[i]) for i in range(len(problem.K1))]
    ccuts = balanced_cone_split(w, nranks)
    feas = _separable_cone_cuts(A, kb, problem.K2)
    if any(c not in feas for c in ccuts[1:-1]):
        # the balanced cuts fall inside a diagonal block (blocks of several K1 cones, uneven blocks): cut at the block boundaries nearest to
        # the balanced targets instead
        if len(feas) < nranks - 1:
            raise ValueError("A has %d diagonal blocks conformal with the K1 / K2 cones, %d ranks asked for: the rows of one rank would reach "
                             "into the columns of another -- A is not block diagonal conformally with the cones" % (len(feas) + 1, nranks))
        cum = np.concatenate([[0.0], np.cumsum(w)])
        fs = sorted(feas)
        ccuts, lo_i = [0], 0
        for g in range(1, nranks):
            target = cum[-1] * g / nranks
            cand = fs[lo_i:len(fs) - (nranks - 1 - g)]               # leave enough block boundaries for the ranks behind
            j = min(range(len(cand)), key=lambda q: abs(cum[cand[q]] - target))
            ccuts.append(int(cand[j]))
            lo_i += j + 1
        ccuts.append(len(problem.K1))
    row_cuts = [int(kb[c]) for c in ccuts]
    # columns touched by each row range
    lo, hi = [], []
    for g in range(nranks):
        sub = A[row_cuts[g]:row_cuts[g + 1]]
        if sub.nnz:
            lo.append(int(sub.indices.min()))
            hi.append(int(sub.indices.max()) + 1)
        else:
            lo.append(None)
            hi.append(None)
    prev_hi = 0
    for g in range(nranks):
        if lo[g] is not None:
            if lo[g] < prev_hi:
                raise ValueError("rows of rank %d reach into the columns of rank %d: A is not block diagonal "
                                 "conformally with the K1 cones" % (g, g - 1))
            prev_hi = hi[g]
    # rank g's columns start at the first column touched by rank g (or a later rank); untouched columns between
    # two blocks go to the left neighbour
    col_cuts = [0]
    for g in range(1, nranks):
        nxt = next((lo[k] for k in range(g, nranks) if lo[k] is not None), n)
        col_cuts.append(max(nxt, col_cuts[-1]))
    col_cuts.append(n)
    return ccuts, row_cuts, col_cuts


def shard_problem(problem: ConicProblem, nranks: int, rank: int) -> Shard:
    ccuts, row_cuts, col_cuts = plan(problem, nranks)
    r0, r1 = row_cuts[rank], row_cuts[rank + 1]
    c0, c1 = col_cuts[rank], col_cuts[rank + 1]
    A = problem.A.tocsr()[r0:r1].tocsc()[:, c0:c1].tocsc()
    A.sort_indices()
    K1 = list(problem.K1[ccuts[rank]:ccuts[rank + 1]])
    K2 = _split_cones_at(problem.K2, col_cuts)[rank]
    loc = ConicProblem("%s[shard %d/%d]" % (problem.name, rank, nranks), A, problem.b[r0:r1].copy(),
                       problem.c[c0:c1].copy(), K1, K2,
                       x0=None if problem.x0 is None else problem.x0[c0:c1].copy(),
                       y0=None if problem.y0 is None else problem.y0[r0:r1].copy(),
                       s0=None if problem.s0 is None else problem.s0[r0:r1].copy(),
                       meta=dict(problem.meta))
    return Shard(loc, (r0, r1), (c0, c1), problem.m, problem.n)


@dataclass
class RowShard:
    problem: ConicProblem          # local problem: the rows of A (and b, K1 cones) this rank owns, ALL columns (c, K2 whole)
    rows: tuple                    # (r0, r1) global row range
    m_global: int
    n: int


def plan_rows(problem: ConicProblem, nranks: int):
    """Contiguous split of the K1 cones into nranks runs of near-equal work (non-zeros + rows); any A qualifies."""
    A = problem.A.tocsr()
    kb = _cone_bounds(problem.K1)
    if len(problem.K1) < nranks:
        raise ValueError("only %d K1 cones for %d ranks" % (len(problem.K1), nranks))
    nnz_per_row = np.diff(A.indptr)
    w = [float(nnz_per_row[kb[i]:kb[i + 1]].sum()) + 4.0 * (kb[i + 1] - kb[i]) for i in range(len(problem.K1))]
    ccuts = balanced_cone_split(w, nranks)
    return ccuts, [int(kb[c]) for c in ccuts]


def shard_rows(problem: ConicProblem, nranks: int, rank: int) -> RowShard:
    """Row sharding for a NON block-diagonal A: rank g keeps the rows of its K1 cones and every column; x, r, c,
    tau, kappa are replicated, y, s, b are local.  Per Q apply the n-vector A'y = sum_g A_g'y_g is all-reduced
    (HSDEAffine.jl:51); the scalar sums count the replicated entries once."""
    ccuts, row_cuts = plan_rows(problem, nranks)
    r0, r1 = row_cuts[rank], row_cuts[rank + 1]
    A = problem.A.tocsr()[r0:r1].tocsc()
    A.sort_indices()
    loc = ConicProblem("%s[rows %d/%d]" % (problem.name, rank, nranks), A, problem.b[r0:r1].copy(), problem.c.copy(),
                       list(problem.K1[ccuts[rank]:ccuts[rank + 1]]), list(problem.K2),
                       x0=None if problem.x0 is None else problem.x0.copy(),
                       y0=None if problem.y0 is None else problem.y0[r0:r1].copy(),
                       s0=None if problem.s0 is None else problem.s0[r0:r1].copy(), meta=dict(problem.meta))
    return RowShard(loc, (r0, r1), problem.m, problem.n)


def rows_local_to_global(shards_z, shards):
    """Global iterate from row-sharded local iterates (x, r, tau, kappa from rank 0: they are replicated)."""
    mg, n = shards[0].m_global, shards[0].n
    lg = mg + n + 1
    z = np.zeros(2 * lg)
    z0 = shards_z[0]
    l0 = shards[0].problem.m + n + 1
    z[:n] = z0[:n]
    z[lg:lg + n] = z0[l0:l0 + n]
    z[lg - 1] = z0[l0 - 1]
    z[2 * lg - 1] = z0[2 * l0 - 1]
    for zl, sh in zip(shards_z, shards):
        r0, r1 = sh.rows
        ml = r1 - r0
        ll = ml + n + 1
        z[n + r0:n + r1] = zl[n:n + ml]
        z[lg + n + r0:lg + n + r1] = zl[ll + n:ll + n + ml]
    return z


def rows_global_to_local(z, shard: RowShard):
    mg, n = shard.m_global, shard.n
    lg = mg + n + 1
    r0, r1 = shard.rows
    return np.concatenate([z[:n], z[n + r0:n + r1], [z[lg - 1]], z[lg:lg + n], z[lg + n + r0:lg + n + r1], [z[2 * lg - 1]]])


def local_to_global(shards_z, shards):
    """Assemble the global iterate z = [x; y; tau; r; s; kappa] from per-rank local iterates (tau, kappa from rank 0)."""
    mg, ng = shards[0].m_global, shards[0].n_global
    lg = mg + ng + 1
    z = np.zeros(2 * lg)
    for zl, sh in zip(shards_z, shards):
        (r0, r1), (c0, c1) = sh.rows, sh.cols
        ml, nl = r1 - r0, c1 - c0
        ll = ml + nl + 1
        z[c0:c1] = zl[0:nl]
        z[ng + r0:ng + r1] = zl[nl:nl + ml]
        z[lg + c0:lg + c1] = zl[ll:ll + nl]
        z[lg + ng + r0:lg + ng + r1] = zl[ll + nl:ll + nl + ml]
    z[lg - 1] = shards_z[0][shards[0].problem.m + shards[0].problem.n]
    z[2 * lg - 1] = shards_z[0][-1]
    return z


def global_to_local(z, shard: Shard):
    mg, ng = shard.m_global, shard.n_global
    lg = mg + ng + 1
    (r0, r1), (c0, c1) = shard.rows, shard.cols
    return np.concatenate([z[c0:c1], z[ng + r0:ng + r1], [z[lg - 1]],
                           z[lg + c0:lg + c1], z[lg + ng + r0:lg + ng + r1], [z[2 * lg - 1]]])
