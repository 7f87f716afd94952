import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check (still part of -m 'not gpu')")
    config.addinivalue_line("markers", "report: prints a measurement or rehearses bench.py over RCCL, asserts little: "
                                       "runs only with GMX_LONG_TESTS=1 (keeps `-m gpu` inside its time limit)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("GMX_LONG_TESTS"):
        return
    skip = pytest.mark.skip(reason="a report, not a check: GMX_LONG_TESTS=1 runs it")
    for item in items:
        if "report" in item.keywords:
            item.add_marker(skip)


def _have_gpu():
    try:
        import gmix_amd
        return gmix_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The product library with a device present; GPU tests fail (not skip) without it."""
    import gmix_amd
    n = gmix_amd.device_count()
    assert n > 0, "no MI355X visible: -m gpu tests must run on the GPU box"
    return gmix_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import gmxo
    gmxo.lib()
    return gmxo
