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


def test_interpreter_selftest():
    """The interpreter itself against results stated by hand (tests/host_emul/selftest/selftest.cpp): every DPP control the product kernels
    use (quad_perm, row_shl / shr / ror, row_bcast:15 / :31, mirror, row and bank masks, bound_ctrl), shuffles, ballot / readlane /
    readfirstlane / any / all with part of the wave returned, barrier + LDS, integer and float atomics from 200 workgroups -- run
    sequentially, on 8 OS threads and in a shuffled order -- and the two aborts: lanes of one wave at different wave operations, and
    a barrier that can never complete."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
    subprocess.check_call(["make", "-s", "-C", here, "selftest/selftest.bin"])
    exe = os.path.join(here, "selftest", "selftest.bin")
    for extra in ({}, {"PLMPM_EMUL_THREADS": "8"}, {"PLMPM_EMUL_SHUFFLE": "5"}):
        p = subprocess.run([exe], env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        assert p.returncode == 0 and p.stdout.decode().strip().endswith("ok"), p.stdout.decode()[-2000:]
    for mode, needle in (("diverge", "divergent collective"), ("deadlock", "deadlock")):
        p = subprocess.run([exe, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        assert p.returncode != 0 and needle in p.stdout.decode(), (mode, p.returncode, p.stdout.decode()[-500:])
