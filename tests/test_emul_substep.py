"""CPU tier: the -m gpu parity tests of the substep, run on the DEVICE SOURCE through the CPU interpreter (tests/emul_engine.py,
tests/host_emul/hipemu) -- the kernels' tiling, in-wave sort, segmented DPP reductions, block flags and the launch logic of the
C ABI, checked against the oracle where no GPU exists.  Same test bodies, same tolerances as tests/test_gpu_substep.py."""
import pytest

import tests.test_gpu_substep as G
from tests import emul_engine
from tests.test_gpu_substep import rolled  # noqa: F401  (fixture)


@pytest.fixture(autouse=True)
def on_interpreter(monkeypatch):
    monkeypatch.setattr(G, "engine_for", emul_engine.engine_for)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_substep_forward_and_adjoint(rolled, dtype):  # noqa: F811
    G.test_substep_forward_and_adjoint(rolled, dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_mass_and_momentum_conservation(rolled, dtype):  # noqa: F811
    G.test_mass_and_momentum_conservation(rolled, dtype)
