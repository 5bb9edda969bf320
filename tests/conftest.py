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
