import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the product library (hipcc cross-compiles gfx950 without a GPU) and the oracle when a
    fresh checkout has neither; both are git-ignored build artefacts."""
    from multi_robot_slam_separators_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    from oracle import pyoracle
    pyoracle.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
