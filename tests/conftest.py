import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_LAUNCHER = None


def _get_launcher():
    global _LAUNCHER
    if _LAUNCHER is None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from launcher import Launcher
        _LAUNCHER = Launcher()
    return _LAUNCHER


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The helper that starts child processes for the tests must exist before anything in this process touches the GPU
    # (tests/launcher.py says why) -- whatever selects the GPU tests (-m gpu, a file name, a plain `pytest tests`), so it
    # is started in every session: it is one idle python process that never imports torch.
    _get_launcher()
    # A GPU session needs its per-test time limit (a kernel that never finishes must fail a test, not hold the box):
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr and not config.pluginmanager.hasplugin("timeout"):
        raise pytest.UsageError("GPU test sessions need pytest-timeout (the per-test limit of tests/conftest.py)")


def pytest_unconfigure(config):
    if _LAUNCHER is not None:
        _LAUNCHER.close()


@pytest.fixture(scope="session")
def launcher():
    """tests/launcher.py client: run child processes without forking this (GPU-initialised) process."""
    assert _LAUNCHER is not None, "the launcher is created in pytest_configure, before any GPU use -- never lazily"
    return _LAUNCHER


def pytest_collection_modifyitems(config, items):
    # A kernel that never finishes must not hold the whole run: every test gets a generous wall-clock limit
    # (pytest-timeout; pytest_configure refuses a GPU session without it; the longest test here takes seconds).
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch
