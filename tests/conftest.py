import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PKG_NAME = "super-lattice-boltzmann-2d_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def slb():
    """The ctypes binding of libslb2d.so (the package name is not an identifier)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def solver_bin():
    path = os.path.join(ROOT, "bin", "boltzmann_solver")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ROOT, "bin/boltzmann_solver"], check=True)
    return path
