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
    # Slab runs started by the tests carry a stall watchdog: a rank of the C loop that makes no
    # progress for 60 s writes down what it was waiting for (host/slb_slabs.c, dump_state) and
    # ends with status 86 -- before any outer time limit kills it without a trace.
    os.environ.setdefault("SLB_WATCHDOG", "60")


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


@pytest.fixture(autouse=True)
def _library_state_does_not_leak(request):
    """After a GPU test the library launches on the null stream again, with default tuning
    and exact sums, and torch's current stream is its default one: a test that leaves another
    stream current would let later tests' copies race with the library's launches."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    mod = importlib.import_module(PKG_NAME)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    L = mod.lib()
    L.slb_set_stream(None)
    L.slb_set_tuning(0, 0, 0)
    L.slb_set_av_mode(mod.SLB_AV_EXACT)
