import os
import sys

import numpy as np
import pytest

os.environ.setdefault("PACX_AUTOBUILD", "1")       # the suite may rebuild a stale libpacx.so (audio-codec_amd/_lib.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tables():
    return np.load(os.path.join(GOLDEN, "tables.npz"))


@pytest.fixture(scope="session")
def stages():
    return np.load(os.path.join(GOLDEN, "stages.npz"))


def load_excerpt(name):
    return np.load(os.path.join(GOLDEN, f"excerpt_{name}.npz"))


EXCERPTS = ["castanet", "harpsichord", "quar48_1", "spmg"]
