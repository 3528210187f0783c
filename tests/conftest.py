import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def split_sd(arrs, prefix, device="cpu"):
    """{'GL.h_net1...': ndarray} -> {'h_net1...': tensor} for one prefix."""
    n = len(prefix)
    return {k[n:]: torch.from_numpy(np.asarray(v)).to(device) for k, v in arrs.items()
            if k.startswith(prefix) and np.asarray(v).dtype.kind in "fiub"}


@pytest.fixture(scope="session")
def ops_small():
    return load_npz("ops_small.npz")


@pytest.fixture(scope="session")
def nets_small():
    return load_npz("nets_small.npz")


@pytest.fixture(scope="session")
def damsm_golden():
    return load_npz("damsm.npz")


@pytest.fixture(scope="session")
def face_c1():
    return load_npz("face_S8_c1.npz")


@pytest.fixture(scope="session")
def face_weights():
    return load_npz("face_S8_weights.npz")
