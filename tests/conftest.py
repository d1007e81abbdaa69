import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/oracle.c) -- the checker, never the thing under test on GPU runs."""
    from oracle import oracle

    return oracle()


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_WIDTHS = [5, 7, 9, 12, 17, 21]
GOLDEN_SIZES = [12, 13, 509, 1000, 4101]


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return {w: np.load(os.path.join(GOLDEN_DIR, f"ref_w{w}.npz")) for w in GOLDEN_WIDTHS}
