import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("RWKV7_CHECK_MASK_HINT", "1")   # backbone: every mark_all_ones / attention_mask_all_ones hint is checked against the mask


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """npz -> dict of torch tensors; uint16 arrays are bf16 bit patterns."""
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype == np.uint16:
            out[k] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads librwkv7_hip.so; GPU tests call through it."""
    from rwkvtts_amd import build, _lib
    build.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co


@pytest.fixture(scope="session")
def lab():
    """tools/lab.py over rwkvtts_amd/lib/librwkv7_hip_lab.so (python -m rwkvtts_amd.build --lab): the superseded A/B twins of the shipped
    kernels.  Cases that cross-check against them are skipped when the lab library has not been built."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lab as _lab
    if not _lab.available():
        pytest.skip("lab library not built (python -m rwkvtts_amd.build --lab)")
    _lab.lib()
    return _lab
