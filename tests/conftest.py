import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")
TSPLIB = os.path.join(GOLDEN, "tsplib")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip and not "0 selected": the product has no CPU fallback.
    expr = (config.getoption("-m") or "").strip()
    wants_gpu = "gpu" in expr and "not gpu" not in expr
    if wants_gpu and any(it.get_closest_marker("gpu") for it in items) and not _has_gpu():
        raise pytest.UsageError("-m gpu was asked for, but no MI355X is visible here (torch.cuda.is_available() is False): "
                                "libteeline_gpu has no CPU fallback, so these tests cannot run — use gpurun")


@pytest.fixture(scope="session")
def tsplib_dir():
    return TSPLIB


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """One tl_ctx for the GPU tests (loads the in-tree libteeline_gpu.so; raises without a GPU)."""
    import teeline_amd
    teeline_amd._capi.load()
    c = teeline_amd.Context(0)
    yield c
    c.close()
