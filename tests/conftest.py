import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun); everything else runs on CPU")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def sub(d, prefix):
    p = prefix + "/"
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


def rel_l1(a, b):
    """Relative L1 error  sum|a-b| / sum|b|  (the north-star tolerance metric: <= 1e-3)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.abs(b).sum()
    return float(np.abs(a - b).sum() / den) if den > 0 else float(np.abs(a - b).sum())


@pytest.fixture(scope="session")
def golden():
    return load_golden
