"""
direct = true on the HIP path (HSDE.jl:12-15, used by the reference's test/testDRandGAPA.jl:37-43 and
test/testprint.jl:49-61): S1 = IndAffine([Q -I], 0) as an exact projection through (I + Q Q')^-1 formed once on the device.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import fos_oracle as orc

pytestmark = pytest.mark.gpu


def _omodel(prob):
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    return orc.Model(prob.A, prob.b, prob.c, codes(prob.K1), codes(prob.K2))


def scrambled_block_problem(pkg, seed=5):
    """A block-diagonal program whose rows and columns are randomly permuted: the diagonal blocks of I + A'A are then NOT contiguous column
    ranges (blocks of 1 ... 40 columns, among them single columns and an empty column)."""
    rng = np.random.default_rng(seed)
    blocks = []
    for s in (1, 40, 7, 1, 23, 12, 3):
        mk = int(rng.integers(s, 4 * s + 6))
        blocks.append(sp.random(mk, s, density=float(rng.choice([0.3, 1.0])), format="csc", random_state=rng, data_rvs=rng.standard_normal))
    blocks.append(sp.csc_matrix((3, 1)))                      # a column without entries: a block of its own (G = 1)
    A = sp.block_diag(blocks, format="csc")
    m, n = A.shape
    A = A[rng.permutation(m)][:, rng.permutation(n)].tocsc()
    A.sort_indices()
    x0 = np.abs(rng.standard_normal(n))
    s0 = np.abs(rng.standard_normal(m))
    y0 = np.zeros(m)
    b = A @ x0 + s0
    c = rng.standard_normal(n)
    return pkg.workloads.ConicProblem("scrambled-blocks", A, b, c, [("NonNeg", m)], [("NonNeg", n)], x0=x0, y0=y0, s0=s0)


@pytest.mark.parametrize("which,mode,form", [("small_mixed", "auto", "block"), ("small_lp", "auto", "block"), ("tile_lp", "auto", "dense"),
                                             ("small_mixed", "dense", "dense"), ("small_lp", "dense", "dense"),
                                             ("block_sdp", "auto", "block"), ("block_sdp", "dense", "dense"), ("scrambled", "auto", "block")])
def test_direct_projection_matches_oracle_and_is_exact(pkg, which, mode, form, monkeypatch):
    """fos_prox_affine in direct mode vs the oracle's IndAffineDirect (1e-12), feasibility Q u = v and orthogonality of the
    displacement to rounding; on a row-block operator, one with odd l, one stored as dual tiles, a block SDP and a scrambled block program --
    in the form fos_enable_direct chooses (the block form wherever I + A'A has diagonal blocks of at most 64 columns: three sweeps per
    projection) and in the dense-inverse form."""
    prob = {"small_mixed": pkg.workloads.small_mixed, "small_lp": lambda: pkg.workloads.small_lp(seed=3, m=31, n=61),
            "tile_lp": lambda: pkg.workloads.small_lp(seed=21, m=96, n=180),
            "block_sdp": lambda: pkg.workloads.c4_block_sdp(nblocks=8, k=16, p=6), "scrambled": lambda: scrambled_block_problem(pkg)}[which]()
    if mode != "auto":
        monkeypatch.setenv("FOS_DIRECT_MODE", mode)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.enable_direct(prob.A)
    assert d.direct_mode() == form
    S1 = orc.IndAffineDirect(orc.HSDEMatrixQ(prob.A, prob.b, prob.c))
    rng = np.random.default_rng(9)
    l = d.l
    for scale in (1.0, 1e3):
        x = scale * rng.standard_normal(d.N)
        y = d.prox_affine(x)
        ref = np.empty(d.N)
        S1.prox(ref, x)
        assert np.linalg.norm(y - ref) <= 1e-12 * np.linalg.norm(ref) * max(1.0, np.linalg.cond(np.eye(l) + S1.Qd @ S1.Qd.T) / 1e3)
        assert np.linalg.norm(d.q_apply(y[:l]) - y[l:]) <= 1e-12 * np.linalg.norm(y)
        u = rng.standard_normal(l)
        t = np.concatenate([u, d.q_apply(u)])
        assert abs((x - y) @ t) <= 1e-11 * np.linalg.norm(t) * np.linalg.norm(x)
    # the CG path of the same handle gives the same projection (to its tolerance floor) once direct is switched off
    d.disable_direct()
    d.reset_affine()
    x = rng.standard_normal(d.N)
    ycg = None
    for _ in range(400):                                  # advance the tolerance schedule to its floor
        ycg = d.prox_affine(x)
    ref = np.empty(d.N)
    S1.prox(ref, x)
    assert np.linalg.norm(ycg - ref) <= 1e-9 * np.linalg.norm(ref)
    d.close()


@pytest.mark.parametrize("algname", ["DR", "GAPA", "FISTA"])
def test_direct_whole_solve_matches_oracle(pkg, algname):
    """Whole solves with direct=true against the oracle's direct mode: there is no inexact CG in the loop, so the iterates do not
    diverge chaotically -- same status, SAME iteration count, residuals and solution to 1e-9; printed table without the cg
    column and history without :cgiter (HSDEStatus.jl:44-50,79; test/testprint.jl:16,56)."""
    prob = pkg.workloads.small_mixed()
    mk = {"DR": lambda M, **o: M.DR(**o), "GAPA": lambda M, **o: M.GAPA(0.8, 0.5, **o), "FISTA": lambda M, **o: M.FISTA(**o)}[algname]
    opts = dict(eps=1e-6, verbose=1, max_iters=3000 if algname != "FISTA" else 300, checki=50, direct=True)
    out = []
    model = pkg.solve(prob, mk(pkg, **opts), out=out)
    oout = []
    sol = orc.solve(_omodel(prob), mk(orc, **opts), out=oout)
    assert out[2] == " Iter | pri res | dua res | rel gap | pri obj | dua obj | kap/tau | time" == oout[2]
    assert "cgiter" not in model.history
    assert model.status() == sol.status and model.iterations == sol.iterations
    last, olast = model.status_obj.last, sol.status_obj.last
    for key in ("p", "d", "g"):
        assert getattr(last, key) == pytest.approx(olast[key], rel=1e-6, abs=1e-12)
    assert np.max(np.abs(model.getsolution() - sol.x)) <= 1e-9 * max(1.0, np.max(np.abs(sol.x)))
    # rows of the table agree in every printed digit except the time column
    rows = [ln for ln in out if ln[:6].strip().isdigit()]
    orows = [ln for ln in oout if ln[:6].strip().isdigit()]
    assert len(rows) == len(orows) and all(a.rsplit(" ", 1)[0] == b.rsplit(" ", 1)[0] for a, b in zip(rows[:3], orows[:3]))


def test_direct_on_dense_lp_is_faster_than_cg_and_reaches_the_optimum(pkg):
    """A 300 x 600 dense LP (dual tiles): direct=true and the CG path reach the same optimum; the direct iteration needs no CG."""
    prob = pkg.workloads.c2_lp(m=300, n=600, scale=25.0)
    opts = dict(eps=1e-6, verbose=0, max_iters=6000, checki=100)
    md = pkg.solve(prob, pkg.DR(direct=True, **opts))
    mi = pkg.solve(prob, pkg.DR(**opts))
    assert md.status() == mi.status()
    assert md.getobjval() == pytest.approx(mi.getobjval(), rel=1e-4, abs=1e-6)
    assert md.getobjval() == pytest.approx(float(prob.c @ prob.x0), rel=1e-2, abs=1e-4)


def test_block_form_refused_or_passed_over_when_not_separable(pkg, monkeypatch):
    """A dense 96 x 180 LP couples all 180 columns: no block form (FOS_DIRECT_MODE=block is refused), the default falls through to the dense inverse."""
    prob = pkg.workloads.small_lp(seed=21, m=96, n=180)
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    monkeypatch.setenv("FOS_DIRECT_MODE", "block")
    with pytest.raises(pkg.lib.FosError):
        d.enable_direct(prob.A)
    assert d.direct_mode() == "off"
    monkeypatch.delenv("FOS_DIRECT_MODE")
    d.enable_direct(prob.A)
    assert d.direct_mode() == "dense"
    d.close()


def _projection_certificate(prob, d, x, tol):
    """y = P(x) onto {(u, v): Q u = v} is the unique point with (i) Q u+ = v+ and (ii) x - y in the row space of [Q -I]: with w = v+ - v,
    u+ - u = Q w (Q' = -Q).  Checked with the oracle's operator (one scipy SpMV pair each) -- no factorisation needed at any size."""
    Q = orc.HSDEMatrixQ(prob.A, prob.b, prob.c)
    l = d.l
    y = d.prox_affine(x)
    t = np.empty(l)
    Q.mul(t, y[:l])
    assert np.linalg.norm(t - y[l:]) <= tol * np.linalg.norm(y), np.linalg.norm(t - y[l:]) / np.linalg.norm(y)
    Q.mul(t, y[l:] - x[l:])
    assert np.linalg.norm((y[:l] - x[:l]) - t) <= tol * np.linalg.norm(x), np.linalg.norm((y[:l] - x[:l]) - t) / np.linalg.norm(x)
    return y


def test_block_form_at_c4_sizes_certificate_cg_and_a_whole_solve(pkg, monkeypatch, fullsize):
    """The 64-block shard of C4 and C4 itself (l = 1 081 345; 512 diagonal blocks of 32 columns): the block form's projection carries the
    oracle-free certificate of the exact projection at 1e-12, equals the warm-started CG at its tolerance floor (FOS_DIRECT_MODE=cg) to 1e-9,
    costs no CG iteration -- and DR(direct=true) solves C4 to its known optimum."""
    for nblk, rng_seed in ((64, 1), (512, 2)):
        prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, nblk))
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.enable_direct(prob.A)
        assert d.direct_mode() == "block"
        rng = np.random.default_rng(rng_seed)
        ys = []
        for scale in (1.0, 1e3):
            x = scale * rng.standard_normal(d.N)
            ys.append((x, _projection_certificate(prob, d, x, 1e-12)))
            assert d.cgiter() == 0
        # idempotent: a point of the set stays
        y2 = d.prox_affine(ys[0][1])
        assert np.linalg.norm(y2 - ys[0][1]) <= 1e-12 * np.linalg.norm(ys[0][1])
        d.close()
        if nblk == 64:
            monkeypatch.setenv("FOS_DIRECT_MODE", "cg")
            dc = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
            dc.enable_direct(prob.A)
            assert dc.direct_mode() == "cg"
            x, y = ys[0]
            ycg = dc.prox_affine(x)
            assert dc.cgiter() > 0 and np.linalg.norm(ycg - y) <= 1e-9 * np.linalg.norm(y)
            dc.close()
            monkeypatch.delenv("FOS_DIRECT_MODE")
    prob = fullsize("C4")
    model = pkg.solve(prob, pkg.DR(direct=True, eps=1e-4, max_iters=4000, verbose=0, checki=250))
    assert model.status() == "Optimal" and "cgiter" not in model.history
    assert model.getobjval() == pytest.approx(float(prob.c @ prob.x0), rel=5e-3)
    assert np.max(np.abs(model.getsolution() - prob.x0)) < 1e-3


def test_block_form_whole_solves_match_the_oracle(pkg):
    """DR / GAPA / FISTA / Dykstra with direct=true on a block SDP (8 x PSD(16), block form on the device, dense Cholesky in the oracle): no inexact
    CG in the loop -- same status, same iteration count, solution to 1e-9."""
    prob = pkg.workloads.c4_block_sdp(nblocks=8, k=16, p=6)
    for mk in (lambda M, **o: M.DR(**o), lambda M, **o: M.GAPA(0.8, 0.5, **o), lambda M, **o: M.FISTA(**o), lambda M, **o: M.Dykstra(**o)):
        opts = dict(eps=1e-6, verbose=0, max_iters=600, checki=50, direct=True)
        model = pkg.solve(prob, mk(pkg, **opts))
        sol = orc.solve(_omodel(prob), mk(orc, **opts), out=[])
        assert model.data.direct_mode() == "block"
        assert model.status() == sol.status and model.iterations == sol.iterations
        assert np.max(np.abs(model.getsolution() - sol.x)) <= 1e-9 * max(1.0, np.max(np.abs(sol.x)))


def test_direct_beyond_the_dense_size_is_the_same_projection_by_cg(pkg, monkeypatch, fullsize):
    """l > 46 000 (here forced: FOS_DIRECT_DENSE_MAX = 10): direct = true keeps its meaning -- the EXACT projection onto {Q u = v} from the
    first call on -- computed by the warm-started CG at its tolerance floor instead of the 0.2^sqrt(i) schedule.  Against the oracle's
    IndAffineDirect from the FIRST call (the scheduled CG is five orders of magnitude off there), and certified to 1e-12 on C3 and C5 themselves."""
    monkeypatch.setenv("FOS_DIRECT_DENSE_MAX", "10")
    monkeypatch.setenv("FOS_DIRECT_MODE", "dense")         # (small_mixed is block separable: the block form would be taken first)
    prob = pkg.workloads.small_mixed()
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.enable_direct(prob.A)
    S1 = orc.IndAffineDirect(orc.HSDEMatrixQ(prob.A, prob.b, prob.c))
    rng = np.random.default_rng(4)
    for k in range(3):
        x = rng.standard_normal(d.N) * (1.0 if k < 2 else 1e3)
        ref = np.empty(d.N)
        S1.prox(ref, x)
        y = d.prox_affine(x)
        assert np.linalg.norm(y - ref) <= 1e-9 * np.linalg.norm(ref), k
        assert d.cgiter() > 0
    # first iterations of a solve follow the oracle's direct = true solve
    oalg = orc.DR(direct=True)
    mo = _omodel(prob)
    oalg.init(mo)
    xo = orc.hsde_initialvalue(mo)
    st = orc.HSDEStatus(mo, 10 ** 9, 1e-9, 0, 1, S1=oalg.S1)
    d.set_alg(pkg.DR(direct=True))
    d.set_iterate(None)
    d.reset_affine()
    for i in range(1, 11):
        st.i = i
        oalg.step(xo, i, st)
        d.step(i, 1, 10 ** 9, 1e-9)
        assert np.linalg.norm(d.get_iterate() - xo) <= 1e-8 * max(1.0, np.linalg.norm(xo)), i
    d.close()
    monkeypatch.delenv("FOS_DIRECT_DENSE_MAX")
    monkeypatch.delenv("FOS_DIRECT_MODE")
    # l = 70 001 (C3) and l = 999 761 (C5): a dense inverse is past the limit, random sparse columns couple everything (no block form) -> CG at the
    # floor.  The projection is certified WITHOUT a factorisation: y is on the set (Q y_u = y_v) and the displacement is in the range of [Q -I]'
    # (x_u - y_u = Q'w with w = y_v - x_v); sigma_min([Q -I]) >= 1, so the distance to the exact projection is at most the sum of the two defects.
    for name in ("C3", "C5"):
        prob = fullsize(name)
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.enable_direct(prob.A)
        assert d.direct_mode() == "cg"
        x = np.random.default_rng(1).standard_normal(d.N)
        l = d.l
        for rep in range(2):                                  # the second one warm-started
            y = d.prox_affine(x)
            on_set = np.linalg.norm(d.q_apply(y[:l]) - y[l:])
            in_range = np.linalg.norm((x[:l] - y[:l]) - d.q_apply(y[l:] - x[l:], transpose=True))
            assert on_set + in_range <= 1e-12 * np.linalg.norm(x), (name, rep, on_set, in_range)
            x = x + 1e-3 * np.random.default_rng(2).standard_normal(d.N)
        d.close()


def test_block_form_on_a_one_rank_communicator_equals_the_single_handle(pkg):
    """direct = true in the SHARDED code path of the block form (the records of the prep and combine kernels reduced locally, all-reduced in stream, the multipliers and
    the tau row formed from the reduced buffer) with a one-rank RCCL communicator: the same sums in the same order as the single handle -- iterates equal to rounding
    (the reduce kernel adds the records in another tree than the kernels that fold them: 1e-13), no CG iteration counted, the mode reported as the block form."""
    prob = pkg.workloads.c4_block_sdp(nblocks=8, k=16, p=6)
    outs = []
    for use_comm in (False, True):
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        if use_comm:
            d.comm_init(1, 0, pkg.HipHSDE.comm_unique_id())
        d.enable_direct(prob.A)
        assert d.direct_mode() == "block"
        d.set_alg(pkg.DR())
        d.set_iterate(None)
        done, checked, res = d.step(1, 40, 40, 1e-6)
        assert done == 40 and checked and d.cgiter() == 0
        outs.append((d.get_iterate(), res.p, res.d, res.g))
        d.close()
    x0, x1 = outs[0][0], outs[1][0]
    assert np.linalg.norm(x1 - x0) <= 1e-12 * max(1.0, np.linalg.norm(x0)), np.linalg.norm(x1 - x0)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert a == pytest.approx(b, rel=1e-9, abs=1e-14)
