import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "mkl_golden.npz"))


@pytest.fixture(scope="session")
def golden_tr():
    """MKL mkl_csrcsc / mkl_scsrmm('T') vectors (tests/golden/make_golden_csrcsc.py)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "mkl_golden_csrcsc.npz"))


@pytest.fixture(scope="session")
def dev():
    """torch device for the HIP path; the product library must be loadable and see a GPU."""
    import torch
    import bofhip
    bofhip.require_device()
    assert torch.cuda.is_available()
    return torch.device("cuda:0")
