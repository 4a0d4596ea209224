import os
import sys
from pathlib import Path

import pytest

# the 16-bit operators look their development switches ($DGA_B16_PLAN, ...) up per call only in a process that says so, once, before
# the library's first launch (csrc/dga_b16.hip b16_dev_env): the tests flip those switches between cases
os.environ.setdefault("DGA_B16_DEV", "1")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dga():
    import deepgemm_ascend_amd as d
    d.build()
    d.lib()
    return d
