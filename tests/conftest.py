import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: timing bounds on a real MI355X (run with -m perf; NOT part of -m gpu, so a noisy lease "
                                       "cannot redden the parity suite; skipped without a GPU)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (halo2-aes_amd/) with every native artefact built."""
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = pkg.Context(0)
    yield c
    c.close()
