import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda", 0)


def pytest_collection_finish(session):
    """a GPU session that runs the whole full-size file: start drawing its seeded weight sets now (tests/test_gpu_fullsize.py)"""
    items = [it for it in session.items if it.fspath.basename == "test_gpu_fullsize.py"]
    if len(items) < 8:
        return
    try:
        import torch
        if torch.cuda.is_available():
            items[0].module.start_synth_prefetch()
    except Exception as e:  # (never fail a collection over a prefetch)
        print(f"[conftest] weight prefetch not started: {e}")
