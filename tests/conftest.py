import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PLMPM_TEST_INTERPRETER=1: run `-m gpu` tests on the CPU interpreter of the device source (tests/emul_engine.py) instead of a GPU --
# what tests/test_emul_tier.py does, case by case, in the CPU-only tier.  A test-side switch only: the engines the tests build are
# replaced here, in the test process; the package itself knows nothing of it.
ON_INTERPRETER = os.environ.get("PLMPM_TEST_INTERPRETER") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if ON_INTERPRETER:
        from tests import emul_engine, gpu_util
        import plasticinelab_amd.engine.core as core
        import plasticinelab_amd.engine.mpm_simulator as ms
        gpu_util.engine_for = emul_engine.engine_for           # (before the test modules import the name)
        ms.Engine = core.Engine = emul_engine.HostEngine       # engines built by the simulator, and by tests that build their own


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are the parity tests proper and need a HIP device plus the in-tree libplmpm.so: without either
    they are skipped with the reason spelled out (a plain `pytest` on a CPU-only box is then green, not 40 errors).
    PLB_REQUIRE_GPU=1 (the GPU box) turns the skip into a failure, so a missing extension cannot pass silently."""
    if ON_INTERPRETER or not any("gpu" in it.keywords for it in items):
        return
    reason = None
    try:
        import torch
        if not torch.cuda.is_available():
            reason = "no HIP device visible"
    except Exception as e:                                   # noqa: BLE001
        reason = f"torch unavailable: {e}"
    if reason is None and not os.path.exists(os.path.join(ROOT, "plasticinelab_amd", "libplmpm.so")):
        reason = "plasticinelab_amd/libplmpm.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    if reason is None:
        return
    if os.environ.get("PLB_REQUIRE_GPU") == "1":
        raise pytest.UsageError(f"PLB_REQUIRE_GPU=1 but {reason}")
    skip = pytest.mark.skip(reason=reason)
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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
