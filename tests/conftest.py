import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_c():
    """Build (if needed) and load the C part of the oracle."""
    import ctypes
    import subprocess
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "libplb_oracle_c.so")
    if not os.path.exists(so) or os.path.getmtime(os.path.join(d, "sdf_sweep.c")) > os.path.getmtime(so):
        subprocess.check_call(["make", "-C", d, "-s"])
    return ctypes.CDLL(so)
