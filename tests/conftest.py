import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Two committed sets of vectors from the reference's own functions (tests/golden/gen_golden.py): the original one and one with
# every input seed shifted by 1000 (WL_GOLDEN_SEED_OFFSET) -- every test that takes `golden` runs against both.
# WL_GOLDEN_DIR: vectors regenerated elsewhere (test_oracle_matches_the_reference_on_fresh_seeds) replace them.
GOLDEN_SETS = ([os.environ["WL_GOLDEN_DIR"]] if os.environ.get("WL_GOLDEN_DIR")
               else [os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests", "golden_seed1000")])
GOLDEN = GOLDEN_SETS[0]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", params=GOLDEN_SETS, ids=[os.path.basename(d.rstrip("/")) for d in GOLDEN_SETS])
def golden(request):
    def load(name):
        return dict(np.load(os.path.join(request.param, name + ".npz")))
    return load
