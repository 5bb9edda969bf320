"""
Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle needs minutes per iteration
at these sizes): linearity / symmetry / skew-symmetry of the operators, idempotence and orthogonality of the
projections, feasibility of the affine projection, and convergence of whole solves to the KNOWN optimum of the
complementary-pair construction (workloads.py).  fp64; tolerances stated per check.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _operator_properties(d, rng, tol=1e-11):
    l, N = d.l, d.N
    x, y = rng.standard_normal(N), rng.standard_normal(N)
    a, b = 0.7, -1.3
    Mx, My = d.kkt_apply(x), d.kkt_apply(y)
    # linearity of [I Q'; Q -I]
    assert np.linalg.norm(d.kkt_apply(a * x + b * y) - (a * Mx + b * My)) <= tol * (np.linalg.norm(Mx) + np.linalg.norm(My))
    # symmetry: x'My == y'Mx   (affinepluslinear.jl:52: transpose == self)
    assert abs(x @ My - y @ Mx) <= tol * np.linalg.norm(x) * np.linalg.norm(My)
    # Q is skew symmetric: u'Qu = 0 and Q' = -Q   (HSDEAffine.jl:61-65)
    u = rng.standard_normal(l)
    Qu = d.q_apply(u)
    assert abs(u @ Qu) <= tol * np.linalg.norm(u) * np.linalg.norm(Qu)
    assert np.array_equal(d.q_apply(u, transpose=True), -Qu)
    # KKT apply is consistent with Q apply:  M [x1;x2] = [x1 - Q x2 ; Q x1 - x2]
    x1, x2 = x[:l], x[l:]
    ref = np.concatenate([x1 - d.q_apply(x2), d.q_apply(x1) - x2])
    assert np.linalg.norm(Mx - ref) <= tol * np.linalg.norm(ref)


def _operator_vs_oracle(prob, d, rng, tol=1e-12):
    """The operators at full size against the oracle directly: mul!(Y, Q, B), mul!(Y, transpose(Q), B) (HSDEAffine.jl:41-65) and
    mul!(y, KKTMatrix(Q), x) (affinepluslinear.jl:37-52) are one scipy SpMV pair each -- as test/HSDEAffine.jl:26-62 does against
    the explicit matrix.  Catches what the properties above cannot: a defect that is linear and hits A and A' alike."""
    import fos_oracle as orc
    Q = orc.HSDEMatrixQ(prob.A, prob.b, prob.c)
    u = rng.standard_normal(d.l)
    ref = np.empty(d.l)
    Q.mul(ref, u)
    got = d.q_apply(u)
    assert np.linalg.norm(got - ref) <= tol * np.linalg.norm(ref)
    assert np.max(np.abs(got - ref)) <= 50 * tol * np.max(np.abs(ref))          # no single row is off (a dropped tile, a short window)
    Q.mul_t(ref, u)
    assert np.linalg.norm(d.q_apply(u, transpose=True) - ref) <= tol * np.linalg.norm(ref)
    x = rng.standard_normal(d.N)
    ref2 = np.empty(d.N)
    orc.KKTMatrix(Q).mul(ref2, x)
    got2 = d.kkt_apply(x)
    assert np.linalg.norm(got2 - ref2) <= tol * np.linalg.norm(ref2)
    assert np.max(np.abs(got2 - ref2)) <= 50 * tol * np.max(np.abs(ref2))
    # structured inputs: unit-like vectors excite single columns / the tau column alone
    e = np.zeros(d.l)
    e[d.l - 1] = 1.0
    Q.mul(ref, e)
    assert np.linalg.norm(d.q_apply(e) - ref) <= tol * max(1.0, np.linalg.norm(ref))
    e = np.zeros(d.l)
    e[::997] = 1.0
    Q.mul(ref, e)
    assert np.linalg.norm(d.q_apply(e) - ref) <= tol * max(1.0, np.linalg.norm(ref))


def _same_step_vs_oracle(pkg, prob, alg, oalg, warm, tol, cg_variant=None):
    """ONE outer iteration of the device and of the oracle from the same steady-state point (iterate, CG warm start, call counter,
    GAPA's alpha12 handed over): solverwrapper.jl:23-29 -> step -> prox!(S1) (CG to the tolerance floor) -> prox!(S2) -> relaxations."""
    import fos_oracle as orc
    BIG = 10 ** 12
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    if cg_variant is not None:
        d.set_cg_variant(cg_variant)
        assert d.cg_variant_name() == cg_variant
    d.set_alg(alg)
    d.set_iterate(None)
    done, _, _ = d.step(1, warm, BIG, 1e-8)
    assert done == warm
    z = d.get_iterate()
    xinit, pi, _ = d.get_affine_state()
    om = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1], [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    oalg.init(om)
    oalg.S1.cgdata.xinit[:] = xinit
    oalg.S1.cgdata.firstrun = False
    oalg.S1.i = pi
    a, b, t, a12 = d.get_alg_state()                 # the algorithm's own *Data struct (fos_get_alg_state)
    if isinstance(oalg, orc.GAPA):
        oalg.alpha12 = a12
    elif isinstance(oalg, orc.FISTA):
        oalg.y[:], oalg.xold[:], oalg.t = a, b, t    # fista.jl:15-25
    elif isinstance(oalg, orc.Dykstra):
        oalg.p[:], oalg.q[:] = a, b                  # dykstra.jl:12-23
    del a, b
    st = orc.HSDEStatus(om, BIG, 1e-8, 0, 0)
    st.i = warm + 1
    xo = z.copy()
    oalg.step(xo, warm + 1, st)
    d.step(warm + 1, 1, BIG, 1e-8)
    zg = d.get_iterate()
    dev = float(np.linalg.norm(zg - xo) / max(1.0, np.linalg.norm(xo)))
    cg_dev, cg_orc = d.cgiter(), oalg.S1.getcgiter()
    d.close()
    print("same-step rel. dev. %.2e, CG iterations %d (oracle %d)" % (dev, cg_dev, cg_orc))
    assert abs(cg_dev - cg_orc) <= 2, (cg_dev, cg_orc)
    assert dev <= tol, dev
    assert np.linalg.norm(zg - z) > 1e3 * tol * max(1.0, np.linalg.norm(z))      # (the step moved the iterate: the comparison is not vacuous)


def _projection_properties(d, rng):
    z = rng.standard_normal(d.N)
    p = d.prox_cones(z)
    pp = d.prox_cones(p)
    assert np.linalg.norm(pp - p) <= 1e-11 * np.linalg.norm(p)             # idempotent
    assert abs((z - p) @ p) <= 1e-10 * np.linalg.norm(z) ** 2               # residual orthogonal to the projection (cone)
    # affine projection: y = [u;v] with Qu = v (to the CG tolerance of the call) and z - y orthogonal to the subspace
    tol = 0.2                                                               # first call of a fresh handle: 0.2^sqrt(1)
    y = d.prox_affine(z)
    u, v = y[:d.l], y[d.l:]
    assert np.linalg.norm(d.q_apply(u) - v) <= 2 * tol
    d.reset_affine()


@pytest.fixture(scope="module")
def c4(pkg, fullsize):
    prob = fullsize("C4")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    yield prob, d
    d.close()


def test_c4_operators_and_projections_full_size(pkg, c4):
    prob, d = c4
    assert (prob.m, prob.n, prob.nnz) == (1064960, 16384, 34078720)
    # the dense 2080 x 32 blocks are held as dual tiles: every entry once (plus the padding lanes of the 32-row tail tiles),
    # all 16384 columns of A finished by the deferred-row kernel
    st = d.operator_stats()
    # (33 tiles of 64 rows per block, stacked into tall tiles of at most FOS_TILE_TALL (default 4) sub-tiles: 9 blocks of 4,4,4,4,4,4,3,3,3)
    assert st["tiles"] % 512 == 0 and 3 <= st["tiles"] // 512 <= 33 and st["tile_vals"] == prob.nnz and st["deferred"] == prob.n
    assert st["slots"] == 32 * st["tiles"]
    assert prob.nnz <= st["vals"] <= 1.05 * prob.nnz and st["cols"] < 0.01 * prob.nnz, st
    rng = np.random.default_rng(0)
    _operator_properties(d, rng)
    _operator_vs_oracle(prob, d, rng)
    _projection_properties(d, rng)


def test_c4_dr_reaches_known_optimum(pkg, c4):
    """512 x PSD(64) block SDP, DR(eps=1e-4): Optimal; objective and x against the known complementary optimum."""
    prob, _ = c4
    model = pkg.solve(prob, pkg.DR(eps=1e-4, max_iters=4000, verbose=0, checki=250))
    assert model.status() == "Optimal"
    opt = float(prob.c @ prob.x0)
    assert model.getobjval() == pytest.approx(opt, rel=5e-3)
    assert np.max(np.abs(model.getsolution() - prob.x0)) < 1e-3
    last = model.status_obj.last
    assert last.p <= 1e-4 * (1 + last.norm_b) and last.d <= 1e-4 * (1 + last.norm_c)
    # primal feasibility of the returned point, checked on the host: b - A x in K1 (PSD blocks) up to the residual
    s = prob.b - prob.A @ model.getsolution()
    k = 64
    dlen = k * (k + 1) // 2
    blk = s[:dlen].copy()
    M = np.zeros((k, k))
    idx = 0
    for j in range(k):
        M[j:, j] = blk[idx:idx + k - j]
        M[j + 1:, j] /= np.sqrt(2)
        idx += k - j
    M = np.tril(M) + np.tril(M, -1).T
    assert np.linalg.eigvalsh(M).min() > -1e-3


def test_c2_full_size_operators_and_progress(pkg, fullsize):
    """Dense 5000 x 10000 LP (all long run-rows): operator properties at full size; DR residuals decrease."""
    prob = fullsize("C2")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    assert prob.nnz == 5000 * 10000
    rng = np.random.default_rng(1)
    _operator_properties(d, rng)
    _operator_vs_oracle(prob, d, rng)
    _projection_properties(d, rng)
    d.set_alg(pkg.DR())
    d.set_iterate(None)
    _, _, r1 = d.step(1, 100, 100, 1e-8)
    _, _, r2 = d.step(101, 300, 400, 1e-8)
    assert r2.p < r1.p and r2.d < r1.d and np.isfinite(r2.g)
    d.close()


def test_c3_full_size_gapa(pkg, fullsize):
    """Sparse SOCP (1000 x SOC(50), nnz ~ 1e6), GAPA: operator properties; reaches Optimal at eps=1e-4 near the known optimum."""
    prob = fullsize("C3")
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    _operator_properties(d, np.random.default_rng(2))
    _operator_vs_oracle(prob, d, np.random.default_rng(12))
    _projection_properties(d, np.random.default_rng(3))
    d.close()
    model = pkg.solve(prob, pkg.GAPA(eps=1e-4, max_iters=6000, verbose=0, checki=250))
    assert model.status() == "Optimal"
    assert model.getobjval() == pytest.approx(float(prob.c @ prob.x0), rel=2e-2, abs=1e-3)


def test_c5_full_size_fista_residuals_vs_oracle(pkg, oracle, fullsize):
    """Mixed-cone HSDE, l ~ 1e6 (NonNeg + SOC + PSD, 8 blocks), FISTA: operator / projection properties at full size, and the
    residuals p, d, g, c'x, b'y the device reports at a check against the oracle's residual formulas (HSDEStatus.jl:27-71) on
    the SAME point -- relative 1e-9 (the BASELINE tolerance is 1e-8)."""
    orc = oracle
    prob = fullsize("C5")
    assert 9.9e5 < prob.m + prob.n + 1 < 1.01e6
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    _operator_properties(d, np.random.default_rng(4))
    _operator_vs_oracle(prob, d, np.random.default_rng(14))
    _projection_properties(d, np.random.default_rng(5))
    d.set_alg(pkg.FISTA())
    d.set_iterate(None)
    done, checked, res = d.step(1, 40, 40, 1e-8)
    assert done == 40 and checked
    z = d.get_checked()                                   # the cone-feasible point the check was evaluated on
    om = orc.Model(prob.A, prob.b, prob.c, [(orc.CONE_CODES[k], l) for k, l in prob.K1],
                   [(orc.CONE_CODES[k], l) for k, l in prob.K2])
    st = orc.HSDEStatus(om, 40, 1e-8, 0, 1)
    st.i = 40
    st.checkstatus(z, override=True)
    for key in ("p", "d", "g", "ctx", "bty"):
        assert getattr(res, key) == pytest.approx(st.last[key], rel=1e-9, abs=1e-13), key
    assert res.status == {"Continue": 0, "Optimal": 1, "Unbounded": 2, "Infeasible": 3}[st.status]
    _, _, res2 = d.step(41, 160, 200, 1e-8)
    assert res2.p < res.p and res2.d < res.d             # and FISTA makes progress at this size
    d.close()


def test_c3_full_size_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """C3 (GAPA), outer iteration 261 from the device's state at iteration 260 (past the CG tolerance floor, ~97 CG iterations):
    the device's iterate against the oracle's, relative 1e-9 (BASELINE tolerance 1e-8)."""
    prob = fullsize("C3")
    _same_step_vs_oracle(pkg, prob, pkg.GAPA(), oracle.GAPA(), 260, 1e-9)


def test_c5_full_size_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """C5 under its own algorithm, FISTA (BASELINE.json configs[4]): the window-panel CG (~140 iterations at the tolerance floor), the cone
    stack (NonNeg + 2000 SOC + 72 PSD(64)) and fista_extrap_kernel as ONE chain -- outer iteration 241 from the device's state at
    iteration 240 (iterate, CG warm start, call counter, and FISTA's y, xold, t through fos_get_alg_state) against the oracle's step
    (fista.jl:28-48), relative 1e-9; then the same operator and cones under DR."""
    prob = fullsize("C5")
    _same_step_vs_oracle(pkg, prob, pkg.FISTA(), oracle.FISTA(), 240, 1e-9)
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 240, 1e-9)


def test_c4_full_size_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """C4 (DR, 512 x PSD(64): dual tiles, 17 CG iterations at the tolerance floor, the batched 1024-matrix PSD projection), outer
    iteration 201 from the device's state at iteration 200 against the oracle's step (solverwrapper.jl:23-29, gap.jl:61-80) --
    relative 1e-9 (BASELINE tolerance 1e-8).  The numpy oracle needs ~2 s for this step."""
    prob = fullsize("C4")
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 200, 1e-9)


def test_c4_raw_full_size_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """C4 exactly as SURVEY 8(d) writes it -- the blocks A_j as drawn, NOT divided by k / 2 (`c4_block_sdp(scale=1)`: bench.py's `value_as_specified`;
    the KKT matrix is worse conditioned, ~35 CG iterations per projection at the tolerance floor instead of 17) -- outer iteration 201 from the device's
    state at iteration 200 against the oracle's step (solverwrapper.jl:23-29, gap.jl:61-80, affinepluslinear.jl:108-118), relative 1e-9."""
    _same_step_vs_oracle(pkg, fullsize("C4raw"), pkg.DR(), oracle.DR(), 200, 1e-9)


def test_c4_shard_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """What one of eight ranks holds of C4 (64 blocks, 128 matrices per projection: the small-batch PSD path -- refinement by matrix
    products from an extrapolated basis, psd64_refine_kernel), outer iterations 201 and 231 against the oracle's step, 1e-9."""
    prob = fullsize("C4shard64")
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 200, 1e-9)
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 230, 1e-9)
    # GAPA: another sequence of matrices for the basis extrapolation (alpha12 changes from step to step)
    _same_step_vs_oracle(pkg, prob, pkg.GAPA(), oracle.GAPA(), 200, 1e-9)


def test_c2_full_size_one_steady_state_iteration_vs_oracle(pkg, oracle, fullsize):
    """C2 (DR, dense 5000 x 10000 LP in tall dual tiles, ~45 CG iterations at the tolerance floor), outer iteration 301 from the
    device's state at iteration 300 against the oracle's step, relative 1e-9.  The numpy oracle needs ~5 s for this step."""
    prob = fullsize("C2")
    _same_step_vs_oracle(pkg, prob, pkg.DR(), oracle.DR(), 300, 1e-9)
