import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def golden_ids(files):
    return [os.path.basename(f)[:-4] for f in files]


def load_golden(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """The parity metric of BASELINE.md section 4: max|a-b| / max|b| per output tensor."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _tuning_back_to_defaults():
    """tgcn_set_tuning switches are process-global: whatever a test set (and however it ended) is undone before the next test runs."""
    yield
    from tgcn_amd import _lib
    if _lib.loaded():
        _lib.lib().tgcn_reset_tuning()
