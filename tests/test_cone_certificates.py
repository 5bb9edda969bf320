"""
The oracle's cone projections against an ORACLE-FREE characterisation (tests/cone_certificates.py): p = P_C(z) iff
p in C, p - z in C*, <p, p - z> = 0.  ProximalOperators.jl's source is not in the reference checkout and no reference test
touches IndSOC / IndRotatedSOC / IndExpPrimal / IndExpDual ("parity unpinned"); these checks pin the
restatements of oracle/fos_oracle.py to the mathematical definition of the sets conemap names (src/cones.jl:4-14) instead.
The same certificate runs on the HIP output at BASELINE sizes in tests/test_gpu_certificates.py.
"""
import numpy as np
import pytest

import fos_oracle as orc
from cone_certificates import DUAL_OF, EXP_RTOL, benign_exp_input, certify_block, certify_stacked_projection, violation

KINDS = ["Free", "Zero", "NonNeg", "NonPos", "SOC", "SOCRotated", "SDP", "ExpPrimal", "ExpDual"]


def _draws(rng, name):
    n = {"SDP": 21, "ExpPrimal": 3, "ExpDual": 3}.get(name, 7)
    for scale in (1e-3, 1.0, 1e3):
        for _ in range(25 if not name.startswith("Exp") else 60):
            z = scale * rng.standard_normal(n)
            if name.startswith("Exp") and not benign_exp_input(z):
                continue
            yield z
    if name in ("SOC", "SOCRotated"):                      # interior, polar, boundary, axis cases
        e = np.zeros(n)
        e[0] = 1.0
        if name == "SOCRotated":
            e[1] = 1.0
        yield e
        yield -e
        yield np.zeros(n)
        v = rng.standard_normal(n)
        v[0] = np.linalg.norm(v[1:])
        yield v
    if name.startswith("Exp"):
        for v in ([1.0, 1.0, np.e], [0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [1.0, -1.0, -1.0], [-2.0, -3.0, 4.0], [5.0, 1e-9, 1.0],
                  [-1.0, 1.0, 0.3], [0.5, 2.0, -1.0], [15.0, 1.0, 1.0], [-15.0, 1.0, 1e-9]):
            yield np.array(v)
            yield -np.array(v)


@pytest.mark.parametrize("name", KINDS)
def test_oracle_projection_is_certified(name):
    """prox! of every cone kind satisfies the three conditions; the Moreau dual (proxDual!, src/cones.jl:80-85) those of C*."""
    rng = np.random.default_rng(100 + KINDS.index(name))
    code = orc.CONE_CODES[name]
    for z in _draws(rng, name):
        p = np.empty_like(z)
        orc.cone_prox(code, p, z)
        certify_block(name, z, p, EXP_RTOL if name.startswith("Exp") else 1e-12)
        orc.cone_prox_dual(code, p, z)
        certify_block(DUAL_OF[name], z, p, EXP_RTOL if name.startswith("Exp") else 1e-12)


@pytest.mark.parametrize("name", ["SOC", "SOCRotated", "SDP", "ExpPrimal", "NonNeg"])
def test_certificate_rejects_wrong_points(name):
    """The certificate is not vacuous: a point moved off the projection (inside the cone, or along the cone) fails one of the
    three conditions."""
    rng = np.random.default_rng(7)
    code = orc.CONE_CODES[name]
    n = {"SDP": 21, "ExpPrimal": 3}.get(name, 7)
    rejected = 0
    for _ in range(40):
        z = rng.standard_normal(n)
        p = np.empty_like(z)
        orc.cone_prox(code, p, z)
        if np.linalg.norm(p - z) < 1e-6 or np.linalg.norm(p) < 1e-6:
            continue                                        # z inside the cone / in the polar: perturbations may stay projections
        for q in (1.001 * p, p + 1e-3 * rng.standard_normal(n), 0.5 * (p + z)):
            try:
                certify_block(name, z, q, EXP_RTOL if name.startswith("Exp") else 1e-9)
            except AssertionError:
                rejected += 1
            else:
                raise AssertionError("certificate accepted a wrong point for %s" % name)
    assert rejected > 30


def test_membership_definitions_on_known_points():
    r2 = np.sqrt(2.0)
    assert violation("SOC", np.array([1.0, 0.6, 0.8])) == pytest.approx(0.0, abs=1e-15)
    assert violation("SOC", np.array([1.0, 0.6, 0.9])) > 0
    assert violation("SOCRotated", np.array([1.0, 2.0, 2.0])) == pytest.approx(0.0, abs=1e-15)      # 2*1*2 = 4 = ||w||^2
    assert violation("SOCRotated", np.array([1.0, 2.0, 2.1])) > 0 and violation("SOCRotated", np.array([-1.0, -2.0, 0.0])) > 0
    assert violation("SDP", np.array([1.0, r2 * 0.5, 1.0])) == pytest.approx(0.0, abs=1e-15)        # [[1,.5],[.5,1]]
    assert violation("SDP", np.array([1.0, r2 * 2.0, 1.0])) == pytest.approx(1.0)                   # eigenvalues -1, 3
    assert violation("ExpPrimal", np.array([1.0, 1.0, np.e])) == pytest.approx(0.0, abs=1e-15)
    assert violation("ExpPrimal", np.array([1.0, 1.0, 2.0])) > 0.1 and violation("ExpPrimal", np.array([-1.0, 0.0, 0.0])) == 0.0
    assert violation("ExpDual", np.array([-1.0, 1.0, np.exp(-2.0)])) == pytest.approx(0.0, abs=1e-15)   # -u exp(v/u) = e^-1 = e w
    assert violation("ExpDual", np.array([0.0, 1.0, 1.0])) == 0.0 and violation("ExpDual", np.array([1.0, 1.0, 1.0])) > 0


def test_stacked_projection_certificate_on_the_oracle():
    """DualConeProduct.prox! (src/cones.jl:122-142) of a problem with every cone kind on both sides."""
    rng = np.random.default_rng(3)
    K1 = [("Zero", 3), ("NonNeg", 4), ("SOC", 5), ("SDP", 10), ("SOCRotated", 4), ("ExpPrimal", 3), ("ExpDual", 3), ("NonPos", 2), ("Free", 2)]
    K2 = [("Free", 3), ("NonNeg", 3), ("SOC", 4), ("ExpPrimal", 3), ("SOCRotated", 5), ("Zero", 2)]
    codes = lambda cs: [(orc.CONE_CODES[k], l) for k, l in cs]
    S2 = orc.DualConeProduct(orc.ConeProduct.from_lengths(codes(K1)), orc.ConeProduct.from_lengths(codes(K2)))
    N = 2 * (sum(l for _, l in K1) + sum(l for _, l in K2) + 1)
    for _ in range(10):
        z = rng.standard_normal(N)
        p = np.empty(N)
        S2.prox(p, z)
        worst = certify_stacked_projection(K1, K2, z, p, rtol=1e-12)
        assert worst.max() < 1e-12
