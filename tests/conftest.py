import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the test infrastructure (oracle, host test lib, corpus generator) once per session.
    The product .so is built too (hipcc cross-compiles without a GPU)."""
    from blingfire_amd import build
    build.build_all()
    yield


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; a plain run on a box without a device skips them
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
