import os
import os.path as osp
import sys

import numpy as np
import pytest

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = osp.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(osp.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
