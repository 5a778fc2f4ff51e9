"""-m gpu: the fused-grid path (grid_op / grid_op.grad evaluated inside the particle kernels' tile fills, PLMPM_FUSE_GRID=1; one GPU
with the per-frame grid store) against the same engine with the grid kernels kept (PLMPM_FUSE_GRID=0).

Both engines evaluate the same per-node arithmetic (mpm_grid.h: grid_node_fwd / grid_node_bwd, after
mpm_simulator.py:189-221), so they must agree to the round-off of the atomic accumulation order: 1e-11 in float64,
1e-5 / 1e-4 (loss / action gradient) in float32, and the fused engine is checked against the oracle's golden rollout
as well.  Also here: the two-particles-per-lane forward kernel (PLMPM_PK=1) against the scalar one."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN
from tests.gpu_util import relerr
from tests.test_gpu_rollout import make_env_sub, run_forward

pytestmark = pytest.mark.gpu


def _experimental_build():
    from plasticinelab_amd import _lib
    return bool(_lib.load().plmpm_build_flags() & 1)


# The two engine variants tested here lost their measurements in round 3 and are compiled only into the experimental build
# of the library (`make -C plasticinelab_amd/csrc experimental`; __graft_entry__.build() builds it too).  The default
# libplmpm.so ignores PLMPM_FUSE_GRID / PLMPM_PK, so against it these tests would compare an engine with itself:
#     PLMPM_LIB=plasticinelab_amd/libplmpm_experimental.so python -m pytest tests/test_gpu_fused_grid.py -m gpu
@pytest.fixture(autouse=True)
def _needs_experimental_build():
    if not _experimental_build():
        pytest.skip("opt-in engine variants: run with PLMPM_LIB=plasticinelab_amd/libplmpm_experimental.so")


class fuse_grid:
    """engines created inside this block are fused-grid engines (on=True) or keep k_grid_op / k_grid_op_grad (on=False);
    the library reads the variable in plmpm_create"""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("PLMPM_FUSE_GRID")
        os.environ["PLMPM_FUSE_GRID"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["PLMPM_FUSE_GRID"]
        else:
            os.environ["PLMPM_FUSE_GRID"] = self.old


def launches(env):
    return {k: v[1] for k, v in env.simulator.engine.profile_read().items() if v[1]}


@pytest.mark.parametrize("tag", ["small", "small_soft"])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_grid_equals_grid_kernels(tag, dtype):
    g = np.load(os.path.join(GOLDEN, f"rollout_{tag}.npz"))
    n, soft = int(g["n_particles"]), bool(g["soft_contact"])
    with fuse_grid(False):
        ref = make_env_sub("Move", n, dtype, soft_contact=soft)
    with fuse_grid(True):
        fus = make_env_sub("Move", n, dtype, soft_contact=soft)
    out = []
    for env in (ref, fus):
        env.simulator.engine.profile_enable(True)
        s0 = env.get_state()["state"]
        loss, grad = run_forward(env, g["actions"], s0)
        fr = env.simulator.engine.get_frame(env.simulator.cur)
        # a second sweep on the same engine: every grid the first one touched must have been left clean
        loss2, grad2 = run_forward(env, g["actions"], s0)
        out.append((loss, grad, fr, loss2, grad2, launches(env)))
        env.simulator.engine.profile_enable(False)
    (l0, g0, f0, l0b, g0b, k0), (l1, g1, f1, l1b, g1b, k1) = out
    print(f"\n[{tag} {dtype}] grid kernels: {k0}\n fused: {k1}\n loss rel {abs(l1 - l0) / abs(l0):.2e} grad rel {relerr(g1, g0):.2e}")
    # the fused engine launched no grid kernel, the reference engine did
    assert "grid_op" in k0 and "grid_op_grad" in k0
    assert "grid_op" not in k1 and "grid_op_grad" not in k1 and "gridop+g2p_p2g" in k1 and "gridop_grad+p2g_grad" in k1
    ltol, gtol, xtol = (1e-11, 1e-9, 1e-11) if dtype == "float64" else (1e-5, 1e-4, 2e-5)
    assert abs(l1 - l0) <= ltol * abs(l0)
    assert relerr(g1, g0) < gtol
    assert relerr(f1["x"], f0["x"]) < xtol
    assert abs(l1b - l1) <= ltol * abs(l1) and relerr(g1b, g1) < gtol
    assert abs(l0b - l0) <= ltol * abs(l0) and relerr(g0b, g0) < gtol


def test_fused_grid_partial_sweeps_leave_clean_grids():
    """Reverse sweeps that stop half way, single substeps, and a forward pass over frames whose reverse pass never ran:
    whatever the call order, the fused engine's grids are clean when they are scattered into again."""
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    n = int(g["n_particles"])
    with fuse_grid(False):
        ref = make_env_sub("Move", n, "float64")
    with fuse_grid(True):
        fus = make_env_sub("Move", n, "float64")
    res = []
    for env in (ref, fus):
        eng, sim = env.simulator.engine, env.simulator
        s0 = env.get_state()["state"]
        sub = sim.substeps
        acts = g["actions"]
        # 1. forward two env steps, reverse only the second one, then everything again from the start
        env.set_state(s0, 666.0, False)
        for a in acts[:2]:
            env.step(a)
        eng.grad_begin(2 * sub)
        eng.add_frame_grad(2 * sub, xa=np.ones((eng.n_particles, 3)))
        eng.step_grad(sub, sub, 1)
        # 2. single substeps forward and backward (plmpm_substep / plmpm_substep_grad)
        env.set_state(s0, 666.0, False)
        eng.set_action(0, sub, acts[0])
        for f in range(3):
            eng.substep(f)
        eng.grad_begin(3)
        eng.add_frame_grad(3, va=np.ones((eng.n_particles, 3)))
        eng.substep_grad(2)
        eng.substep_grad(1)                 # frame 0 is left without its reverse substep
        xg = eng.get_frame_grad(1)["x"]
        # 3. the full rollout
        loss, grad = run_forward(env, acts, s0)
        res.append((xg, loss, grad))
    (x0, l0, g0), (x1, l1, g1) = res
    assert relerr(x1, x0) < 1e-9
    assert abs(l1 - l0) <= 1e-11 * abs(l0)
    assert relerr(g1, g0) < 1e-9
    assert abs(l1 - float(g["loss"])) / abs(float(g["loss"])) < 1e-10
    assert relerr(g1, g["grad"]) < 1e-7


class packed_pairs:
    """fp32 engines created inside this block run the two-particles-per-lane forward kernel (plmpm_kernels_pk.h)"""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("PLMPM_PK")
        os.environ["PLMPM_PK"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["PLMPM_PK"]
        else:
            os.environ["PLMPM_PK"] = self.old


@pytest.mark.parametrize("fg", [False, True])
def test_packed_pair_kernel_equals_scalar_kernel(fg):
    """k_g2p_p2g_pk (two particles per lane, packed fp32: mpm_math.h instantiated for P2 / D2 / I2) against the scalar
    kernel on the same rollout: the arithmetic per particle is the same sequence of operations (tests/test_host_emul.py
    checks the two instantiations bit for bit on the host), so only the atomic accumulation order differs."""
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    n = int(g["n_particles"])
    with fuse_grid(fg):
        with packed_pairs(False):
            ref = make_env_sub("Move", n, "float32")
        with packed_pairs(True):
            pk = make_env_sub("Move", n, "float32")
    l0, g0 = run_forward(ref, g["actions"])
    l1, g1 = run_forward(pk, g["actions"])
    f0, f1 = ref.simulator.engine.get_frame(ref.simulator.cur), pk.simulator.engine.get_frame(pk.simulator.cur)
    print(f"\n[pk fg={fg}] loss rel {abs(l1 - l0) / abs(l0):.2e} grad rel {relerr(g1, g0):.2e} x {relerr(f1['x'], f0['x']):.2e}")
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert relerr(g1, g0) < 1e-4
    assert relerr(f1["x"], f0["x"]) < 2e-6 and relerr(f1["F"], f0["F"]) < 2e-5
    assert abs(l1 - float(g["loss"])) / abs(float(g["loss"])) < 1e-5 and relerr(g1, g["grad"]) < 1e-4
