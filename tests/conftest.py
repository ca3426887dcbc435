import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never be silently skipped on a GPU box; on a CPU-only box
    # they are deselected by `-m "not gpu"`.  If somebody runs them without a
    # GPU, fail loudly rather than pass on a fallback.
    pass
