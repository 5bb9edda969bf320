"""
CPU: closed-form optima of textbook conic programs (tests/analytic_cases.py) reached by the oracle's DR -- an oracle-independent pin of the
cone conventions (entry order, the 2pq of the rotated cone, the e of the dual exponential cone, svec's sqrt 2) for the cones no reference
test touches.  The HIP path runs the same cases in tests/test_gpu_known_answers.py.
"""
import numpy as np
import pytest

import fos_oracle as orc
from analytic_cases import cases

CASES = cases()


@pytest.mark.parametrize("cs", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reaches_the_closed_form_optimum(cs):
    model = orc.Model(cs["A"], cs["b"], cs["c"], [(orc.CONE_CODES[k], l) for k, l in cs["K1"]], [(orc.CONE_CODES[k], l) for k, l in cs["K2"]])
    sol = orc.solve(model, orc.DR(eps=1e-8, verbose=0, max_iters=20000))
    assert sol.status == "Optimal"
    assert abs(sol.obj_val - cs["opt"]) < 1e-8 * max(1.0, abs(cs["opt"]))
    assert np.abs(sol.x - cs["x"]).max() < 1e-7
    for w in cs["wrong"]:                                  # what a differently ordered / scaled cone would give is far away
        assert abs(sol.obj_val - w) > 1e-3


def test_cases_are_what_they_claim():
    """The closed forms themselves: x* is feasible for the stated cone definitions, attains the stated value, and a first-order
    perturbation inside the feasible set does not improve it (cheap sanity of the case table, no solver involved)."""
    import math
    for cs in CASES:
        s = cs["b"] - cs["A"] @ cs["x"]
        assert float(cs["c"] @ cs["x"]) == pytest.approx(cs["opt"], rel=1e-14, abs=1e-14), cs["name"]
        off = 0
        for kind, l in cs["K1"]:
            _member(kind, s[off:off + l], cs["name"])
            off += l
        off = 0
        for kind, l in cs["K2"]:
            _member(kind, cs["x"][off:off + l], cs["name"])
            off += l


def _member(kind, v, name, tol=1e-12):
    import math
    if kind == "Free":
        return
    if kind == "Zero":
        assert np.abs(v).max() <= tol, name
    elif kind == "NonNeg":
        assert v.min() >= -tol, name
    elif kind == "SOC":
        assert np.linalg.norm(v[1:]) <= v[0] + tol, name
    elif kind == "SOCRotated":
        assert v[0] >= -tol and v[1] >= -tol and v[2:] @ v[2:] <= 2 * v[0] * v[1] + tol, name
    elif kind == "ExpPrimal":
        assert v[1] > 0 and v[1] * math.exp(v[0] / v[1]) <= v[2] + tol, name
    elif kind == "ExpDual":
        assert v[0] < 0 and -v[0] * math.exp(v[1] / v[0]) <= math.e * v[2] + tol, name
    elif kind == "SDP":
        k = int(round((math.sqrt(8 * len(v) + 1) - 1) / 2))
        M = np.zeros((k, k))
        idx = 0
        for j in range(k):
            for i in range(j, k):
                M[i, j] = M[j, i] = v[idx] if i == j else v[idx] / math.sqrt(2.0)
                idx += 1
        assert np.linalg.eigvalsh(M).min() >= -tol, name
    else:
        raise AssertionError(kind)
