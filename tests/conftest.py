import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    # No single test may hold a run hostage (a GPU box is charged by the minute; a multi-process case whose ranks the box does not
    # schedule would otherwise sit in a rendezvous until the caller's own limit): 300 s per test unless the command line says otherwise.
    # The slowest legitimate case (the Kimi-K2-shaped oracle pass on the box's 16-CPU quota) takes ~40 s.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = float(os.environ.get("KTX_TEST_TIMEOUT", "300"))


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()
