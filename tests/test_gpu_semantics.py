"""-m gpu: the two unverified pieces of Taichi autodiff semantics are switches in the engine
(plmpm_config.contact_min_adjoint / minmax_tie) and in the oracle (plb_oracle.SEMANTICS); engine and oracle agree under
either setting, and the settings really differ on inputs that exercise them -- so pinning them against Taichi-generated
vectors (tests/golden/make_taichi_golden.py) is a flag flip on both sides, not a kernel edit."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests.util import O, ROOT, oracle_scene, sparse_target
from tests.gpu_util import engine_for, load_state, preroll, relerr
from tests.test_gpu_loss import c_sdf

pytestmark = pytest.mark.gpu


class semantics:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = dict(O.SEMANTICS)
        O.SEMANTICS.update(self.kw)

    def __exit__(self, *a):
        O.SEMANTICS.update(self.old)


@pytest.fixture(scope="module")
def oracle_c():
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "libplb_oracle_c.so"))


def test_hard_contact_min_adjoint_add_and_argmin(oracle_c):
    """ti.atomic_min(min_dist, d) in the hard contact loss (loss.py:123-128): its adjoint to every particle ("add", the
    assumed Taichi 0.7.x behaviour) or to the particle attaining the minimum ("argmin")."""
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=2000)
    acts = np.zeros((1, 6)); acts[:, 0] = 0.3; acts[:, 3] = -0.2
    state, mats, poses = preroll(sim, prims, x0, acts)
    # lift the manipulators clear of the body: with a particle inside one, min_dist = 0 and its adjoint 2 min_dist vanishes
    poses = [(p + torch.tensor([0.0, 0.12, 0.0], dtype=O.DT), r) for p, r in poses]
    tgt = sparse_target("Move3D-v1")
    ref_sdf = c_sdf(oracle_c, tgt, sim.dx)
    got = {}
    for mode in ("add", "argmin"):
        eng = engine_for(sim, prims, dtype="float64", contact_min_adjoint=mode)
        load_state(eng, 0, state, mats, poses)
        eng.loss_set_target(tgt)
        eng.loss_set_weights(0, 0, 1, False)                      # the contact term alone
        out = eng.loss_forward(0)
        x = state[0].clone().requires_grad_(True)
        pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
        with semantics(contact_min_adjoint=mode):
            L, parts = O.compute_loss(sim, O.LossCfg(soft_contact=False), prims, x, pin,
                                      torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(ref_sdf.reshape(-1)))
            Lc = parts["contact_loss"]
            gs = torch.autograd.grad(Lc, [x] + [p for p, _ in pin], allow_unused=True)
        assert abs(out["contact_loss"] - float(Lc)) <= 1e-10 * abs(float(Lc))
        eng.grad_begin(0)
        eng.loss_backward(0)
        gx = eng.get_frame_grad(0)["x"]
        assert relerr(gx, gs[0].numpy()) < 1e-9, mode
        for k in range(len(prims)):
            ref = np.zeros(3) if gs[1 + k] is None else gs[1 + k].numpy()
            assert np.abs(eng.get_primitive_grad(k, 0)[:3] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-12) + 1e-12, mode
        got[mode] = gx
        eng.close()
    # "add" reaches every particle outside a manipulator, "argmin" one particle per manipulator
    assert (np.abs(got["add"]).sum(1) > 0).sum() > 1000
    assert 1 <= (np.abs(got["argmin"]).sum(1) > 0).sum() <= len(prims)


@pytest.mark.parametrize("tie", ["second", "first"])
def test_minmax_tie_routing_at_a_clamped_manipulator(tie):
    """position[f+1] = max(min(position[f] + v[f], upper), lower) (primive_base.py:119) with the manipulator sitting exactly
    ON its upper bound and not moving in x: min(y, upper) is an exact tie.  Its adjoint goes to `upper` ("second": no
    gradient reaches the x action) or to y ("first": it does).  Engine (k_fk_chain_grad) against the oracle's autograd."""
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=500)
    p0 = prims[0]
    prims[0] = O.PrimCfg(**{**p0.__dict__, "upper_bound": (p0.init_pos[0], 1.0, 1.0)})
    action = np.array([0.0, 0.3, -0.2, 0.1, 0.0, 0.2])
    sub = sim.substeps
    # oracle: the kinematics chain of one env step, seeded with d/d position[last] = (1, 2, 3) per manipulator
    a = torch.tensor(action, dtype=O.DT, requires_grad=True)
    with semantics(minmax_tie=tie):
        total, ofs = 0.0, 0
        for p in prims:
            v, w = O.set_velocity(p, a[ofs:ofs + p.action_dim], sub)[:2]
            ofs += p.action_dim
            pos, rot = torch.tensor(p.init_pos, dtype=O.DT), torch.tensor(p.init_rot, dtype=O.DT)
            for _ in range(sub):
                pos, rot = O.forward_kinematics(p, pos, rot, v, w)[:2]
            total = total + (pos * torch.tensor([1.0, 2.0, 3.0], dtype=O.DT)).sum()
        (g_ref,) = torch.autograd.grad(total, a)
    eng = engine_for(sim, prims, dtype="float64", max_frames=sub + 1, minmax_tie=tie)
    load_state(eng, 0, O.init_state(x0), O.materials(sim), O.init_poses(prims))
    eng.set_action(0, sub, action)
    eng.step(0, sub)
    eng.grad_begin(sub)
    for k in range(len(prims)):
        eng.add_primitive_grad(k, sub, [1.0, 2.0, 3.0, 0, 0, 0, 0])
    eng.step_grad(0, sub, 0)
    g = eng.get_action_grad(1)[0]
    eng.close()
    # the particles are far from the manipulators' path in one env step: only the seeded pose adjoint reaches the actions
    assert np.abs(g - g_ref.numpy()).max() < 1e-12, (g, g_ref)
    assert (g[0] == 0.0) == (tie == "second") and abs(g[1] - 0.02) < 1e-12
