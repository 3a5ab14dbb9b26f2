import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vqtest_util import TRAIN_CASES, load_golden  # noqa: E402

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): C restatement loaded through ctypes."""
    from oracle import vq_oracle
    vq_oracle.build()
    return vq_oracle


@pytest.fixture(params=TRAIN_CASES)
def train_case(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


