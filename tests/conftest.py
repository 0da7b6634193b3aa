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


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Read-only view of one tests/golden/*.npz fixture (arrays -> torch tensors)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __contains__(self, k):
        return k in self.z.files

    def np(self, k):
        return self.z[k]

    def __getitem__(self, k):
        a = self.z[k]
        if a.shape == ():
            return a.item()
        return torch.from_numpy(a)

    def keys(self, prefix=""):
        return [k for k in self.z.files if k.startswith(prefix)]

    def bits(self, k, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(np.unpackbits(self.z[k])[:n].reshape(shape).astype(bool))

    def tape(self, k="noise"):
        n = int(self.z[k + "/n"])
        out = []
        names = {}
        for f in self.z.files:
            if f.startswith(k + "/") and f != k + "/n":
                _, i, kind = f.split("/")
                names[int(i)] = (kind, f)
        for i in range(n):
            kind, f = names[i]
            out.append((kind, torch.from_numpy(self.z[f])))
        return out


@pytest.fixture
def golden():
    return Golden


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{a.numel()} off; max abs err {float(err.max()):.3e}, "
        f"max rel {float((err / b.abs().clip(min=1e-12)).max()):.3e}")
