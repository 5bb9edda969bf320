import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "oracle") not in sys.path:
    sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory `firstordersolvers.jl_amd`, imported as firstordersolvers_jl_amd)."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    import fos_oracle
    return fos_oracle


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


class _ProblemCache:
    """The full-size BASELINE.json problems, built ONCE per test session (C4 and C5 were generated four times each, C2 three times, across
    test_gpu_fullsize / test_gpu_certificates / test_gpu_direct / test_gpu_resident: ~40 s of a 500 s GPU suite)."""

    def __init__(self, pkg):
        self.pkg, self.cache = pkg, {}

    def __call__(self, name):
        if name not in self.cache:
            w = self.pkg.workloads
            self.cache[name] = {"C2": w.c2_lp, "C3": w.c3_socp, "C4": w.c4_block_sdp, "C5": w.c5_mixed,
                                "C4raw": lambda: w.c4_block_sdp(scale=1.0),
                                "C4shard64": lambda: w.c4_block_sdp(nblocks=512, block_range=(0, 64))}[name]()
        return self.cache[name]


@pytest.fixture(scope="session")
def fullsize(pkg):
    return _ProblemCache(pkg)
