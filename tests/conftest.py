import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """GPU tests need the in-tree libemu_hip.so (a git-ignored build product).  If a GPU session starts in a tree where it
    was never built, build it once here (hipcc is part of the image); the product itself still fails loudly without it."""
    if not any("gpu" in item.keywords for item in items):
        return
    from emu_amd import _lib, build
    if os.path.exists(_lib.LIB_PATH):
        return
    try:
        build._hipcc()
    except RuntimeError:
        return                                  # nothing to build with: the tests will report the missing library
    build.build(verbose=False)
