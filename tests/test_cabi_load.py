"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/foship.h declares; the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp


def test_library_exports_every_header_symbol(pkg):
    lib = pkg.lib.load(check_symbols=True)
    declared = pkg.lib.header_symbols()
    assert set(declared) == set(pkg.lib.PROTOTYPES), "binding table and include/foship.h disagree"
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.fos_abi_version() == 1
    assert ctypes.sizeof(pkg.lib.CheckResult) == 11 * 8 + 8 + 4 + 4


def test_no_cpu_fallback(pkg):
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(pkg.lib.FosError) as ei:
        pkg.HipHSDE(sp.csc_matrix(np.eye(2)), np.zeros(2), np.zeros(2), [("Zero", 2)], [("Free", 2)])
    assert ei.value.code == -6          # FOS_ENODEVICE


def test_constructor_signatures_mirror_reference(pkg):
    """gap.jl:13, solvers.jl:10-11, gapa.jl:15, fista.jl:11, dykstra.jl:9 defaults and keyword capture."""
    g = pkg.GAP()
    assert (g.alpha, g.alpha1, g.alpha2, g.direct, g.options) == (0.8, 1.8, 1.8, False, {})
    d = pkg.DR(eps=1e-8, verbose=0)
    assert (d.alpha, d.alpha1, d.alpha2) == (0.5, 2.0, 2.0) and d.options == dict(eps=1e-8, verbose=0)
    a = pkg.AP()
    assert (a.alpha, a.alpha1, a.alpha2) == (1, 1.0, 1.0)
    ga = pkg.GAPA(0.8, 0.9, checki=100)
    assert (ga.alpha, ga.beta, ga.options) == (0.8, 0.9, dict(checki=100))
    assert pkg.GAPA().alpha == 1.0 and pkg.GAPA().beta == 0.0
    assert pkg.FISTA().alpha == 1.0
    assert pkg.Dykstra(max_iters=5).options == dict(max_iters=5)
    m = pkg.FOSMathProgModel(pkg.GAP(0.5, 2.0, 2.0, max_iters=2000))
    assert m.options == dict(max_iters=2000) and m.status() == "NotSolved"
    assert "SDP" in m.supportedcones()
    assert pkg.GAPA(direct=True).direct is True and pkg.DR(0.5, direct=True).direct is True     # gap.jl:10 (device: tests/test_gpu_direct.py)


def test_cone_normalisation(pkg):
    from firstordersolvers_jl_amd.interface import _normalize_cones
    t, s, l = _normalize_cones([("NonNeg", range(1, 4)), ("SOC", [4, 5, 6, 7]), ("Free", 2)], 9, "K1")
    assert list(t) == [2, 4, 0] and list(s) == [1, 4, 8] and list(l) == [3, 4, 2]
    with pytest.raises(ValueError):
        _normalize_cones([("Zero", [1, 3])], 2, "K1")          # toRanges error, cones.jl:50
    with pytest.raises(KeyError):
        _normalize_cones([("Banana", 2)], 2, "K1")
