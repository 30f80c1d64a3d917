import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g

    p = g.load_package()
    if not os.path.exists(p.library_path()):  # fresh checkout: the in-tree .so files are git-ignored
        g.build()
    return p


@pytest.fixture(scope="session")
def O():
    import __graft_entry__ as g

    return g.load_oracle()


GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_path(name):
    return os.path.join(GOLDEN, name)
