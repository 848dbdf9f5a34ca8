import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# WL_GOLDEN_DIR: vectors regenerated elsewhere (test_oracle_matches_the_reference_on_fresh_seeds); default: the committed ones
GOLDEN = os.environ.get("WL_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load
