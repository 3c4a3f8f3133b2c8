import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def th_oracle(oracle):
    """the oracle with the Thompson tables of the default mp_options built (thompson_init)"""
    from icar_amd.options import options_t
    p, f = options_t().mp_options.as_arrays()
    oracle.thompson_init(p, f)
    return oracle
