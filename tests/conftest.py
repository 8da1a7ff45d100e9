import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the product library and the oracle once per session if they are missing
    (in-tree .so files normally travel with the snapshot)."""
    import __graft_entry__ as entry

    entry.build(only_if_missing=True)
