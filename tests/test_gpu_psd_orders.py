"""PSD cones of order above 64 (cones.jl:11,89-94 -> IndPSD(scaling=true)): projected by the matrix sign function on the fp64 matrix cores
(csrc/psd_sign.hip -- P = (M + M sign(M)) / 2, sign by a polynomial iteration of batched products) in both handles: the HSDE path's
prox!(., S2, .) (primal copy s and dual copy y) and the Feasibility form's ConeProduct set.  The oracle is the restated eigendecomposition
(oracle/cones.py).  Cases the iteration could get wrong: spectra across fourteen decades, exact zero eigenvalues, +-lambda pairs, a zero
matrix, orders that are not multiples of the 64-wide tile, orders on either side of the switch in one handle."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _tri(k):
    return k * (k + 1) // 2


def _spectrum_case(pkg, rng, k, kind):
    Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
    if kind == "decades":
        w = np.sign(rng.standard_normal(k)) * 10.0 ** rng.uniform(-14, 0, k)
    elif kind == "rank-deficient":
        w = rng.standard_normal(k); w[k // 3:] = 0.0
    elif kind == "pairs":
        w = np.repeat(rng.uniform(0.5, 2.0, (k + 1) // 2), 2)[:k] * np.tile([1.0, -1.0], (k + 1) // 2)[:k]
    elif kind == "clustered":
        w = np.where(rng.random(k) < 0.5, 1.0, -1.0) * (1.0 + 1e-9 * rng.standard_normal(k))
    else:
        w = rng.standard_normal(k) * 100.0
    return pkg.workloads._svec((Q * w) @ Q.T)


@pytest.mark.parametrize("orders", [(65,), (128,), (200, 96), (64, 70, 5, 129)], ids=lambda o: "-".join(map(str, o)))
def test_hsde_cone_projection_of_large_orders(pkg, oracle, orders):
    orc = oracle
    rng = np.random.default_rng(100 + sum(orders))
    K1 = [("SDP", _tri(k)) for k in orders] + [("NonNeg", 3)]
    m = sum(l for _, l in K1)
    A = sp.random(m, 4, density=0.05, format="csc", random_state=np.random.RandomState(1))
    d = pkg.HipHSDE(A, np.zeros(m), np.zeros(4), K1, [("Free", 4)])
    S2 = orc.DualConeProduct(orc.ConeProduct.from_lengths([(orc.CONE_CODES[c], l) for c, l in K1]), orc.ConeProduct.from_lengths([(orc.CONE_FREE, 4)]))
    l, n = d.l, 4
    cases = [rng.standard_normal(d.N), np.zeros(d.N)]
    for kind in ("decades", "rank-deficient", "pairs", "clustered", "wide"):
        z = 0.1 * rng.standard_normal(d.N)
        off = 0
        for k in orders:
            sv = _spectrum_case(pkg, rng, k, kind)
            z[l + n + off:l + n + off + _tri(k)] = sv                 # the primal copy
            z[n + off:n + off + _tri(k)] = -sv[::1]                   # the dual copy: x + P(-x)
            off += _tri(k)
        cases.append(z)
    ref = np.empty(d.N)
    for i, z in enumerate(cases):
        S2.prox(ref, z)
        out = d.prox_cones(z)
        assert np.linalg.norm(out - ref) <= 5e-13 * max(1.0, np.linalg.norm(z)), (orders, i)
        out2 = d.prox_cones(out)                                      # idempotent: a projected point stays where it is
        assert np.linalg.norm(out2 - out) <= 5e-13 * max(1.0, np.linalg.norm(z)), (orders, i)
    d.close()


def test_feasibility_cone_set_of_large_orders(pkg, oracle):
    orc = oracle
    cones = [("NonNeg", 5), ("SDP", _tri(96)), ("SOC", 4), ("SDP", _tri(10)), ("SDP", _tri(130))]
    n = sum(l for _, l in cones)
    K = orc.ConeProduct.from_lengths([(orc.CONE_CODES[c], l) for c, l in cones])
    d = pkg.HipFeasibility(pkg.Feasibility(pkg.ConeProduct(cones), pkg.IndBox(-np.inf, np.inf), n))
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    y = np.empty(n)
    for rep in range(3):
        K.prox(y, x)
        yd = d.prox(1, x)
        assert np.abs(yd - y).max() <= 5e-12 * max(1.0, np.abs(x).max()), rep
        x = x + 1e-3 * rng.standard_normal(n)


def test_block_sdp_of_order_80_solves_like_the_oracle(pkg, oracle):
    """A whole DR solve through the sign path: a block SDP with three PSD(80) cones, against the oracle's solve of the same problem
    (status, iteration count within one check interval, solution, objective)."""
    from test_gpu_parity import omodel
    orc = oracle
    prob = pkg.workloads.c4_block_sdp(nblocks=3, k=80, p=12, seed=5)
    model = pkg.solve(prob, pkg.DR(eps=1e-4, max_iters=3000, verbose=0), out=[])
    sol = orc.solve(omodel(prob), orc.DR(eps=1e-4, max_iters=3000, verbose=0), out=[])
    assert model.status() == sol.status == "Optimal"
    assert abs(model.iterations - sol.iterations) <= model.options.get("checki", 100)
    x = model.getsolution()
    assert np.max(np.abs(x - sol.x[:len(x)])) <= 1e-7 * max(1.0, np.abs(sol.x).max())
