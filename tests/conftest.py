import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` on CPU."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _sane_cpu_threads():
    """The full-size oracle tests raise torch's intra-op thread count to 64; left in place it makes the thousands of tiny
    CPU ops of the small-model oracle runs that follow crawl (a test took 74 s instead of 1 s on the GPU box's 16-core share).
    Every test starts at 8 threads; the tests that want more set it themselves."""
    try:
        import torch
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    except Exception:  # pragma: no cover
        pass
    yield
