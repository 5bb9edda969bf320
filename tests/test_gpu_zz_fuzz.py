"""A fixed slice of the randomised differential campaign (tests/fuzz_parity.py: random sizes, densities, cone partitions over every cone kind,
operator formats, algorithms, wrappers -- HIP path against the oracle).  The full campaign (round 4: 3 000 HSDE seeds -- products, projections, CG,
whole solves --, 6 600 HSDE seeds with direct = true and 5 000 Feasibility seeds -- 20 iterates each --, 600 large operators) found no discrepancy beyond the situations
the script documents, in which the reference's own arithmetic is decided by rounding noise.
(Named to run LAST under `pytest -x`: a campaign slice is the test most exposed to a one-ulp difference between boxes, and must not hide the others.)"""
import pytest

import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lo", [0, 1000])
def test_hsde_slice(pkg, lo):
    bad = []
    for seed in range(lo, lo + 30):
        tag, fails = fuzz_parity.one_seed(pkg, seed, seed % 5 == 0)
        if fails:
            bad.append((tag, fails))
    assert not bad, bad


def test_feasibility_slice(pkg):
    bad = []
    for seed in range(0, 120):
        tag, fails = fuzz_parity.one_feas_seed(pkg, seed)
        if fails:
            bad.append((tag, fails))
    assert not bad, bad


def test_hsde_direct_iterates_slice(pkg):
    bad = []
    for seed in range(0, 150):
        tag, fails = fuzz_parity.one_direct_seed(pkg, seed)
        if fails:
            bad.append((tag, fails))
    assert not bad, bad


def test_hsde_block_direct_iterates_slice(pkg):
    """direct = true on randomly permuted block-diagonal operators: the device's block form (three sweeps) against the oracle's dense Cholesky, 20 iterates each."""
    bad = []
    for seed in range(0, 100):
        tag, fails = fuzz_parity.one_direct_seed(pkg, seed, blocky=True)
        if fails:
            bad.append((tag, fails))
    assert not bad, bad
