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


TIE_SHAPES = {
    # particles exactly ON the +x face of the box: min(max(q), 0) is the tie min(0, 0) -- "first" lets d sdf/d x = 1 through
    # (the other branch, length(max(q, 0)), has value 0 there and contributes no gradient), "second" stops it at the constant
    "Box": (dict(shape="Box", size=(0.125, 0.0625, 0.125)), lambda n: (0.625, np.linspace(0.45, 0.55, n), np.linspace(0.42, 0.58, n))),
    # on the top cap of the cylinder (axis y; the reference's h is its radius, r its half height): same min(0, 0), seen in d / d y
    "Cylinder": (dict(shape="Cylinder", h=0.125, r=0.0625), lambda n: (np.linspace(0.45, 0.55, n), 0.5625, 0.5)),
    # on the mid-plane between the two sticks: min(sdf_a, sdf_b) of two bitwise-equal distances -- the adjoint goes to stick
    # a ("first") or to stick b ("second"): d sdf/d x changes sign, and so does the adjoint of the gap
    "Chopsticks": (dict(shape="Chopsticks", h=0.25, r=0.03125, init_gap=0.125, minimal_gap=0.0625),
                   lambda n: (0.5, np.linspace(0.30, 0.45, n), np.linspace(0.45, 0.55, n))),
}


@pytest.mark.parametrize("shape", list(TIE_SHAPES))
def test_minmax_tie_in_the_contact_loss_through_shape_sdfs(shape, oracle_c):
    """minmax_tie applies to every max / min on the differentiated path (include/plmpm.h), the shape SDFs under the contact
    loss included (k_loss_grad -> shape_local_adj): particles placed EXACTLY on a tie of the shape's SDF (coordinates that
    are exact in binary), soft contact loss alone, engine against the oracle's autograd under both settings -- and the two
    settings really give different gradients there."""
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=600)
    kw, place = TIE_SHAPES[shape]
    adim = 7 if shape == "Chopsticks" else 6
    prims = [O.PrimCfg(init_pos=(0.5, 0.5, 0.5), init_rot=(1.0, 0.0, 0.0, 0.0), action_dim=adim, action_scale=(0.01,) * adim, **kw)]
    n_tie = 40
    x0 = x0.copy()
    px, py, pz = place(n_tie)
    x0[:n_tie, 0], x0[:n_tie, 1], x0[:n_tie, 2] = px, py, pz
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    tgt = sparse_target("Move3D-v1")
    ref_sdf = c_sdf(oracle_c, tgt, sim.dx)
    got = {}
    for tie in ("second", "first"):
        eng = engine_for(sim, prims, dtype="float64", minmax_tie=tie)
        load_state(eng, 0, state, mats, poses)
        eng.loss_set_target(tgt)
        eng.loss_set_weights(0, 0, 1, True)                       # the soft contact term alone
        out = eng.loss_forward(0)
        x = state[0].clone().requires_grad_(True)
        pin = [tuple(t.clone().requires_grad_(True) for t in po) for po in poses]
        with semantics(minmax_tie=tie):
            L, parts = O.compute_loss(sim, O.LossCfg(soft_contact=True), prims, x, pin,
                                      torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(ref_sdf.reshape(-1)))
            Lc = parts["contact_loss"]
            gs = torch.autograd.grad(Lc, [x] + [t for po in pin for t in po], allow_unused=True)
        assert abs(out["contact_loss"] - float(Lc)) <= 1e-10 * abs(float(Lc))
        eng.grad_begin(0)
        eng.loss_backward(0)
        gx = eng.get_frame_grad(0)["x"]
        assert relerr(gx, gs[0].numpy()) < 1e-9, tie
        ref_pose = np.concatenate([np.zeros(t.numel()) if g is None else g.numpy().reshape(-1) for g, t in zip(gs[1:], pin[0])])
        pg = eng.get_primitive_grad(0, 0)[:len(ref_pose)]
        assert np.abs(pg - ref_pose).max() <= 1e-9 * max(np.abs(ref_pose).max(), 1e-300), (tie, pg, ref_pose)
        got[tie] = gx[:n_tie].copy()
        eng.close()
    # on the tie the two routings differ: in d loss / d x of the particles that sit on it
    c = 1 if shape == "Cylinder" else 0
    assert np.abs(got["first"][:, c] - got["second"][:, c]).max() > 1e-3 * max(np.abs(got["first"]).max(), np.abs(got["second"]).max())
