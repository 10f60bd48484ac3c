import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads libctrlv_hip.so; GPU tests fail loudly if the native library is missing."""
    import __graft_entry__ as g
    g.build()             # no-op when the library's build id matches the hash of csrc/ + include/
    from ctrlv_amd import _lib
    return _lib.load()


@pytest.fixture(autouse=True)
def _grad_mode_is_per_test():
    """A test (or a module it imports) that flips torch's global grad mode must not leak it into the next test."""
    import torch
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    yield
    torch.set_grad_enabled(prev)
