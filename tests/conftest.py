import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liboracle.so) — checker only."""
    from tests import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the whole session (GPU tests only)."""
    import arrow_go_amd as ah
    c = ah.Context(0)
    yield c
    c.close()
