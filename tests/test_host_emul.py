"""CPU tier: the per-particle / per-node arithmetic the HIP kernels are built from (csrc/mpm_math.h,
mpm_grid.h) compiled for the host (tests/host_emul, test infrastructure only) against the oracle:
forward substep and the hand-derived adjoint -- including the closed-form SVD-free constitutive VJP with the
reference's 1e-6 clamp, quaternion pose adjoints and the primitive kinematics chain."""
import os
import numpy as np
import pytest
import torch

from tests import emul
from tests.util import O, oracle_scene


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def case():
    torch.manual_seed(0)
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=1500)
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    acts = torch.zeros(2, 6, dtype=O.DT); acts[:, 0] = 0.9; acts[:, 3] = -0.9; acts[:, 1] = 0.3
    with torch.no_grad():
        for a in acts:
            state, poses = O.env_step(sim, prims, 666.0, state, mats, poses, a)
    vel = [O.set_velocity(p, acts[0][3 * k:3 * k + 3], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, pos, rot, v, w) for p, (pos, rot), (v, w) in zip(prims, poses, vel)]
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
    nin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in nxt]
    out = O.substep(sim, prims, 666.0, sin, mats, pin, nin)
    cot = [torch.randn_like(t) for t in out]
    inputs = list(sin) + [t for pr in pin for t in pr] + [t for pr in nin for t in pr]
    gs = torch.autograd.grad(sum((o * c).sum() for o, c in zip(out, cot)), inputs, allow_unused=True)
    gs = [torch.zeros_like(t) if g is None else g for g, t in zip(gs, inputs)]
    assert (state[3] - torch.eye(3, dtype=O.DT)).abs().max() > 0.1          # the state is really deformed / yielding
    return sim, prims, state, mats, poses, nxt, out, cot, gs


@pytest.mark.parametrize("use_float,tol", [(False, 1e-11), (True, 5e-6)])
def test_substep_math_matches_oracle(case, use_float, tol):
    sim, prims, state, mats, poses, nxt, out, cot, gs = case
    ec = emul.make_cfg(sim, len(prims), 666.0, use_float=use_float)
    pa = emul.make_prims(prims, [(p.numpy(), r.numpy()) for p, r in poses], [(p.numpy(), r.numpy()) for p, r in nxt])
    st, mt = [t.numpy() for t in state], [t.numpy() for t in mats]
    e = emul.substep(ec, pa, st, mt)
    for a, b in zip(e, out):
        assert relerr(a, b.detach().numpy()) < tol
    (xa, va, Ca, Fa), pose = emul.substep_grad(ec, pa, st, mt, out[1].detach().numpy(), [c.numpy() for c in cot])
    for a, b in zip((xa, va, Ca, Fa), gs[:4]):
        assert relerr(a, b.numpy()) < 3 * tol
    P = len(prims)
    for k in range(P):
        ref = np.concatenate([gs[4 + 2 * k].numpy(), gs[5 + 2 * k].numpy(),
                              gs[4 + 2 * P + 2 * k].numpy(), gs[5 + 2 * P + 2 * k].numpy()])
        assert np.abs(ref).max() > 0 and relerr(pose[k][:14], ref) < 3 * tol


def test_kinematics_chain_and_adjoint():
    torch.manual_seed(3)
    p = O.PrimCfg(shape="Sphere", lower_bound=(0.0, 0.0, 0.0), upper_bound=(0.6, 1.0, 1.0), action_dim=6,
                  action_scale=(0.01,) * 6)
    pos = torch.tensor([0.595, 0.3, 0.002], dtype=O.DT, requires_grad=True)
    rot = torch.tensor([0.8, 0.2, -0.5, 0.1], dtype=O.DT); rot = (rot / rot.norm()).requires_grad_(True)
    v = torch.tensor([0.01, -0.004, -0.005], dtype=O.DT, requires_grad=True)      # hits the upper x and lower z clamps
    w = torch.tensor([0.02, -0.01, 0.03], dtype=O.DT, requires_grad=True)
    pos1, rot1 = O.forward_kinematics(p, pos, rot, v, w)
    e_pos1, e_rot1 = emul.fk_fwd(pos.detach().numpy(), rot.detach().numpy(), v.detach().numpy(), w.detach().numpy(),
                                 p.lower_bound, p.upper_bound)
    assert np.allclose(e_pos1, pos1.detach().numpy(), atol=1e-15) and np.allclose(e_rot1, rot1.detach().numpy(), atol=1e-15)
    cp, cr = torch.randn(3, dtype=O.DT), torch.randn(4, dtype=O.DT)
    gs = torch.autograd.grad((pos1 * cp).sum() + (rot1 * cr).sum(), [pos, rot, v, w])
    got = emul.fk_bwd(pos.detach().numpy(), rot.detach().numpy(), v.detach().numpy(), w.detach().numpy(),
                      p.lower_bound, p.upper_bound, cp.numpy(), cr.numpy())
    for a, b in zip(got, gs):
        assert np.allclose(a, b.numpy(), atol=1e-13)
    assert got[2][0] == 0.0 and got[2][2] == 0.0 and got[2][1] != 0.0              # clamp gates the gradient (Q7-like)


def test_static_cylinder_contact_forward(case):
    """Rope's static Cylinder obstacle (h = radius, r = half height): forward parity of collide."""
    cfg, sim, prims, x0 = oracle_scene("Rope", 1, n_particles=1500)
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    a = torch.tensor([0.5, 0.2, -0.9, -0.5, 0.2, -0.9], dtype=O.DT)
    with torch.no_grad():
        state, poses = O.env_step(sim, prims, 666.0, state, mats, poses, a)
        vel = [O.set_velocity(p, a[3 * k:3 * k + 3], sim.substeps) if p.action_dim else (torch.zeros(3, dtype=O.DT),) * 2
               for k, p in enumerate(prims)]
        nxt = [O.forward_kinematics(p, pos, rot, v, w) for p, (pos, rot), (v, w) in zip(prims, poses, vel)]
        out = O.substep(sim, prims, 666.0, state, mats, poses, nxt)
    ec = emul.make_cfg(sim, len(prims), 666.0)
    pa = emul.make_prims(prims, [(p.numpy(), r.numpy()) for p, r in poses], [(p.numpy(), r.numpy()) for p, r in nxt])
    e = emul.substep(ec, pa, [t.numpy() for t in state], [t.numpy() for t in mats])
    for x, y in zip(e, out):
        assert relerr(x, y.numpy()) < 1e-11


@pytest.mark.parametrize("shape,kw", [("Capsule", dict(h=0.06, r=0.03)), ("Torus", dict(tx=0.05, ty=0.02)),
                                      ("Cylinder", dict(h=0.05, r=0.04)), ("Box", dict(size=(0.04, 0.03, 0.05)))])
def test_capsule_torus_pose_adjoints(shape, kw):
    """Movable Capsule (writer.yml) / Torus (torus.yml) -- and Cylinder / Box, which no reference task moves but
    SURVEY 8f rank 2 lists: tilted, 6-dof actions so that position AND rotation adjoints at frames f and f+1 are
    exercised; hand-derived sdf-gradient / normal-Jacobian vs oracle autograd."""
    torch.manual_seed(0)
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=1500)
    rot = np.array([0.9, 0.2, -0.3, 0.25]); rot /= np.linalg.norm(rot)
    prims = [O.PrimCfg(shape=shape, init_pos=p.init_pos, init_rot=tuple(rot), friction=0.9, action_dim=6,
                       action_scale=(0.01,) * 6, **kw) for p in prims]
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    acts = torch.tensor([[0.9, 0.3, 0.1, 0.5, -0.4, 0.3, -0.9, 0.2, -0.1, -0.3, 0.6, 0.2]], dtype=O.DT).repeat(2, 1)
    with torch.no_grad():
        for a in acts:
            state, poses = O.env_step(sim, prims, 666.0, state, mats, poses, a)
    vel = [O.set_velocity(p, acts[0][6 * k:6 * k + 6], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, pos, r, v, w) for p, (pos, r), (v, w) in zip(prims, poses, vel)]
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
    nin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in nxt]
    out = O.substep(sim, prims, 666.0, sin, mats, pin, nin)
    cot = [torch.randn_like(t) for t in out]
    inputs = list(sin) + [t for pr in pin for t in pr] + [t for pr in nin for t in pr]
    gs = torch.autograd.grad(sum((o * c).sum() for o, c in zip(out, cot)), inputs, allow_unused=True)
    gs = [torch.zeros_like(t) if g is None else g for g, t in zip(gs, inputs)]
    ec = emul.make_cfg(sim, len(prims), 666.0)
    pa = emul.make_prims(prims, [(p.numpy(), r.numpy()) for p, r in poses], [(p.numpy(), r.numpy()) for p, r in nxt])
    st, mt = [t.numpy() for t in state], [t.numpy() for t in mats]
    (xa, va, Ca, Fa), pose = emul.substep_grad(ec, pa, st, mt, out[1].detach().numpy(), [c.numpy() for c in cot])
    for a, b in zip((xa, va, Ca, Fa), gs[:4]):
        assert relerr(a, b.numpy()) < 1e-10
    P = len(prims)
    for k in range(P):
        ref = np.concatenate([gs[4 + 2 * k].numpy(), gs[5 + 2 * k].numpy(), gs[4 + 2 * P + 2 * k].numpy(), gs[5 + 2 * P + 2 * k].numpy()])
        assert np.abs(ref[3:7]).max() > 0 and relerr(pose[k][:14], ref) < 1e-10


def test_rollingpin_kinematics_and_adjoint():
    """RollingPin.forward_kinematics (primitives.py:66-80) and its hand-derived adjoint vs oracle autograd."""
    torch.manual_seed(5)
    p = O.PrimCfg(shape="RollingPin", h=0.3, r=0.03, lower_bound=(0.0, 0.0, 0.0), upper_bound=(1.0, 1.0, 1.0), action_dim=3,
                  action_scale=(0.6666666666666667, 0.06666666666666668, 0.001))
    pos = torch.tensor([0.5, 0.123, 0.5], dtype=O.DT, requires_grad=True)
    rot = torch.tensor([0.707, 0.707, 0.1, -0.05], dtype=O.DT); rot = (rot / rot.norm()).requires_grad_(True)
    v = torch.tensor([0.035, -0.0035, -5e-5], dtype=O.DT, requires_grad=True)
    pos1, rot1 = O.forward_kinematics(p, pos, rot, v, torch.zeros(3, dtype=O.DT))
    e_pos1, e_rot1 = emul.fk_rollingpin_fwd(pos.detach().numpy(), rot.detach().numpy(), v.detach().numpy(), p.lower_bound, p.upper_bound)
    assert np.allclose(e_pos1, pos1.detach().numpy(), atol=1e-15) and np.allclose(e_rot1, rot1.detach().numpy(), atol=1e-15)
    cp, cr = torch.randn(3, dtype=O.DT), torch.randn(4, dtype=O.DT)
    gs = torch.autograd.grad((pos1 * cp).sum() + (rot1 * cr).sum(), [pos, rot, v])
    got = emul.fk_rollingpin_bwd(pos.detach().numpy(), rot.detach().numpy(), v.detach().numpy(), p.lower_bound, p.upper_bound,
                                 cp.numpy(), cr.numpy())
    for a, b in zip(got, gs):
        assert np.allclose(a, b.numpy(), rtol=1e-12, atol=1e-13)


def _np_pose(pose):
    return tuple(t.detach().numpy() for t in pose)


def test_chopsticks_contact_and_gap_adjoints():
    """Chopsticks (primitives.py:83-154): double-capsule sdf/normal, body-frame rotation, gap degree of freedom.
    One substep after a short closing rollout; pose adjoints at f and f+1 AND the gap adjoint vs oracle autograd."""
    torch.manual_seed(0)
    cfg, sim, _, x0 = oracle_scene("Move", 1, n_particles=1500)
    rot = np.array([0.95, 0.1, -0.2, 0.15]); rot /= np.linalg.norm(rot)
    prims = [O.PrimCfg(shape="Chopsticks", h=0.2, r=0.02, init_pos=(0.67, 0.74, 0.75), init_rot=tuple(rot), friction=10.0,
                       action_dim=7, action_scale=(0.02, 0.02, 0.02, 0.04, 0.04, 0.04, 0.02), init_gap=0.12,
                       minimal_gap=0.06)]
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    act = torch.tensor([0.2, -0.9, 0.1, 0.3, -0.2, 0.4, 0.5], dtype=O.DT)
    with torch.no_grad():
        for _ in range(2):
            state, poses = O.env_step(sim, prims, 666.0, state, mats, poses, act)
    vel = [O.set_velocity(prims[0], act, sim.substeps)]
    nxt = [O.forward_kinematics(p, po[0], po[1], vw[0], vw[1], po[2], vw[2]) for p, po, vw in zip(prims, poses, vel)]
    assert 0.06 < float(nxt[0][2]) < 0.12
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    pin = [tuple(t.clone().requires_grad_(True) for t in po) for po in poses]
    nin = [tuple(t.clone().requires_grad_(True) for t in po) for po in nxt]
    out = O.substep(sim, prims, 666.0, sin, mats, pin, nin)
    cot = [torch.randn_like(t) for t in out]
    inputs = list(sin) + list(pin[0]) + list(nin[0])
    gs = torch.autograd.grad(sum((o * c).sum() for o, c in zip(out, cot)), inputs, allow_unused=True)
    gs = [torch.zeros_like(t) if g is None else g for g, t in zip(gs, inputs)]
    ec = emul.make_cfg(sim, 1, 666.0)
    pa = emul.make_prims(prims, [_np_pose(p) for p in poses], [_np_pose(p) for p in nxt])
    st, mt = [t.numpy() for t in state], [t.numpy() for t in mats]
    e = emul.substep(ec, pa, st, mt)
    for a, b in zip(e, out):
        assert relerr(a, b.detach().numpy()) < 1e-11
    (xa, va, Ca, Fa), pose = emul.substep_grad(ec, pa, st, mt, out[1].detach().numpy(), [c.numpy() for c in cot])
    for a, b in zip((xa, va, Ca, Fa), gs[:4]):
        assert relerr(a, b.numpy()) < 1e-10
    ref = np.concatenate([gs[4].numpy(), gs[5].numpy(), gs[7].numpy(), gs[8].numpy(), gs[6].numpy().reshape(1)])
    assert abs(ref[14]) > 0 and np.abs(ref[3:7]).max() > 0
    assert relerr(pose[0], ref) < 1e-10


def test_chopsticks_kinematics_and_adjoint():
    """Chopsticks.forward_kinematics (primitives.py:94-98) and its adjoint, on both sides of the minimal-gap clamp."""
    torch.manual_seed(7)
    p = O.PrimCfg(shape="Chopsticks", h=0.2, r=0.02, action_dim=7, action_scale=(0.02,) * 7, minimal_gap=0.06,
                  lower_bound=(0.0, 0.0, 0.0), upper_bound=(1.0, 1.0, 1.0))
    for gap0, gv0 in ((0.1, 0.01), (0.065, 0.01), (0.06, 0.0)):
        pos = torch.tensor([0.5, 0.15, 0.5], dtype=O.DT, requires_grad=True)
        rot = torch.tensor([0.8, 0.3, 0.1, -0.4], dtype=O.DT); rot = (rot / rot.norm()).requires_grad_(True)
        v = torch.tensor([0.001, -0.002, 0.0005], dtype=O.DT, requires_grad=True)
        w = torch.tensor([0.002, -0.001, 0.003], dtype=O.DT, requires_grad=True)
        gap = torch.tensor(gap0, dtype=O.DT, requires_grad=True)
        gv = torch.tensor(gv0, dtype=O.DT, requires_grad=True)
        pos1, rot1, gap1 = O.forward_kinematics(p, pos, rot, v, w, gap, gv)
        args = [t.detach().numpy() for t in (pos, rot, v, w)] + [gap0, gv0, p.minimal_gap, p.lower_bound, p.upper_bound]
        e = emul.fk_chopsticks_fwd(*args)
        assert np.allclose(e[0], pos1.detach().numpy(), atol=1e-15) and np.allclose(e[1], rot1.detach().numpy(), atol=1e-15)
        assert e[2] == float(gap1)
        cp, cr, cg = torch.randn(3, dtype=O.DT), torch.randn(4, dtype=O.DT), torch.randn((), dtype=O.DT)
        gs = torch.autograd.grad((pos1 * cp).sum() + (rot1 * cr).sum() + gap1 * cg, [pos, rot, gap, v, w, gv], allow_unused=True)
        got = emul.fk_chopsticks_bwd(*args, cp.numpy(), cr.numpy(), float(cg))
        for a, b in zip(got, gs):
            b = 0.0 if b is None else b.numpy()
            assert np.allclose(a, b, rtol=1e-12, atol=1e-13)


def test_node_between_two_manipulators():
    """Two overlapping spheres pressed into the same corner of the body: nodes in contact with BOTH exercise the branch
    of grid_node_bwd that recomputes the velocity entering the earlier collide (the later one reuses its forward
    intermediates), in the velocity-only pass and in the pose pass."""
    torch.manual_seed(1)
    cfg, sim, prims0, x0 = oracle_scene("Move", 1, n_particles=1500)
    top = x0[np.argmax(x0[:, 1])]
    c = np.array([top[0], top[1] + 0.025, top[2]])
    prims = [O.PrimCfg(shape="Sphere", radius=0.04, init_pos=tuple(c + d), friction=0.9, action_dim=3, action_scale=(0.01,) * 3)
             for d in (np.array([-0.008, 0.0, 0.0]), np.array([0.008, 0.004, 0.0]))]
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    a = torch.tensor([0.3, -0.8, 0.1, -0.2, -0.9, 0.2], dtype=O.DT)
    vel = [O.set_velocity(p, a[3 * k:3 * k + 3], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, pos, r, v, w) for p, (pos, r), (v, w) in zip(prims, poses, vel)]
    # both spheres must reach the same nodes, or the test checks nothing
    n = sim.n_grid
    gp = torch.stack(torch.meshgrid(*[torch.arange(n, dtype=O.DT) / n] * 3, indexing="ij"), -1).reshape(-1, 3)
    near = [(torch.linalg.norm(gp - pos, dim=1) - 0.04) < 0.0 for pos, _ in poses]
    assert int((near[0] & near[1]).sum()) > 20
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
    nin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in nxt]
    out = O.substep(sim, prims, 666.0, sin, mats, pin, nin)
    cot = [torch.randn_like(t) for t in out]
    inputs = list(sin) + [t for pr in pin for t in pr] + [t for pr in nin for t in pr]
    gs = torch.autograd.grad(sum((o * c).sum() for o, c in zip(out, cot)), inputs, allow_unused=True)
    gs = [torch.zeros_like(t) if g is None else g for g, t in zip(gs, inputs)]
    ec = emul.make_cfg(sim, len(prims), 666.0)
    pa = emul.make_prims(prims, [(p.numpy(), r.numpy()) for p, r in poses], [(p.numpy(), r.numpy()) for p, r in nxt])
    st, mt = [t.numpy() for t in state], [t.numpy() for t in mats]
    (xa, va, Ca, Fa), pose = emul.substep_grad(ec, pa, st, mt, out[1].detach().numpy(), [c.numpy() for c in cot])
    for u, g in zip((xa, va, Ca, Fa), gs[:4]):
        assert relerr(u, g.numpy()) < 1e-10
    P = len(prims)
    for k in range(P):
        ref = np.concatenate([gs[4 + 2 * k].numpy(), gs[5 + 2 * k].numpy(), gs[4 + 2 * P + 2 * k].numpy(), gs[5 + 2 * P + 2 * k].numpy()])
        assert np.abs(ref[:3]).max() > 0 and relerr(pose[k][:14], ref) < 1e-10


def _oracle_constitutive(Et, mu, lam, ys, GS, GF):
    """stress (unscaled) and new_F of p2g + the F_tmp adjoint of <GS, stress> + <GF, new_F> through the oracle's own SVD
    with the reference's literal backward_svd (oracle/plb_oracle.py::SvdRef, compute_von_mises)."""
    Ft = (torch.as_tensor(Et) + torch.eye(3, dtype=O.DT)).requires_grad_(True)
    U, sig, V = O.SvdRef.apply(Ft)
    mu_t, lam_t = torch.full((len(Et),), mu, dtype=O.DT), torch.full((len(Et),), lam, dtype=O.DT)
    newF, yields = O.compute_von_mises(Ft, U, sig, V, torch.full((len(Et),), ys, dtype=O.DT), mu_t)
    J = torch.linalg.det(newF)
    r = U @ V.transpose(-1, -2)
    stress = 2 * mu * (newF - r) @ newF.transpose(-1, -2) + torch.eye(3, dtype=O.DT) * (lam_t * J * (J - 1))[:, None, None]
    (g,) = torch.autograd.grad((stress * torch.as_tensor(GS)).sum() + (newF * torch.as_tensor(GF)).sum(), Ft)
    return stress.detach().numpy(), newF.detach().numpy(), g.numpy(), yields.numpy()


@pytest.mark.parametrize("use_float,tol", [(False, 2e-10), (True, 2e-5)])
def test_elastic_fast_path_matches_svd_path_and_oracle(use_float, tol):
    """Round 4: waves in which no particle can yield skip the SVD (polar rotation by Newton steps, closed-form VJP).  One
    particle = one wave here.  Strains from 1e-7 to ~5 %, with rotations up to ~0.5 rad: the fast path must be TAKEN for
    the clearly elastic ones, never for a yielding one, and agree with the Jacobi path and with the oracle's literal
    SVD + backward_svd (clamp 1e-6) in stress, new_F and the F_tmp adjoint."""
    rng = np.random.default_rng(3)
    n = 4000
    mu, lam, ys = 5000 / 2.4, 5000 * 0.2 / (1.2 * 0.6), 200.0
    scale = 10 ** rng.uniform(-7, -1.3, n)
    strain = rng.standard_normal((n, 3, 3)) * scale[:, None, None]
    strain = 0.5 * (strain + strain.transpose(0, 2, 1))
    w = rng.standard_normal((n, 3)) * rng.uniform(0, 0.5, n)[:, None]
    Wm = np.zeros((n, 3, 3))
    Wm[:, 0, 1], Wm[:, 0, 2], Wm[:, 1, 2] = -w[:, 2], w[:, 1], -w[:, 0]
    Wm -= Wm.transpose(0, 2, 1)
    R = torch.linalg.matrix_exp(torch.as_tensor(Wm)).numpy()
    Et = R @ (np.eye(3) + strain) - np.eye(3)
    Et[:50] = 0.0                                        # the undeformed state: every eigenvalue gap is 0 (clamp regime)
    Et[50:100] = R[50:100] - np.eye(3)                    # pure rotations
    GS, GF = rng.standard_normal((n, 3, 3)), rng.standard_normal((n, 3, 3))
    s_o, F_o, g_o, y_o = _oracle_constitutive(Et, mu, lam, ys, GS, GF)
    s_f, E_f, g_f, fast = emul.constitutive(Et, mu, lam, ys, GS, GF, use_float=use_float, allow_fast=True)
    s_s, E_s, g_s, took = emul.constitutive(Et, mu, lam, ys, GS, GF, use_float=use_float, allow_fast=False)
    assert not took.any()
    assert not (fast & y_o).any()                        # sufficient condition: never fast when the oracle yields
    lam3 = np.linalg.eigvalsh(np.einsum("nki,nkj->nij", Et + np.eye(3), Et + np.eye(3)) - np.eye(3))
    gap = np.minimum(lam3[:, 1] - lam3[:, 0], lam3[:, 2] - lam3[:, 1])
    clear = (~y_o) & (scale < 0.02) & (gap > (1e-4 if use_float else 3e-6))
    assert fast[clear].mean() > 0.98 and fast[:50].all() and fast.mean() > 0.5     # (pure rotations: gaps of round-off size, not 0 -- Jacobi path)
    # relative to each particle's own magnitudes (stress spans seven decades here)
    def err(a, b):
        return np.abs(a - b).max(axis=(1, 2)) / np.maximum(np.abs(b).max(axis=(1, 2)), 1e-300)
    f = fast
    assert (err(E_f[f] + np.eye(3), F_o[f]) < tol).all()
    # A = F^T F - I loses eps x angle^2 to cancellation under a rotation whichever path consumes it: the fast path must be
    # as good as the Jacobi path there, and both within tol of the oracle where that floor is below it
    e_f, e_s = err(s_f[f], s_o[f]), err(s_s[f], s_o[f])
    assert (e_f < 20 * tol + 2 * e_s).all(), (e_f - 2 * e_s).max()
    floor = (6e-8 if use_float else 1e-16) * (w[f] ** 2).sum(1) / scale[f]
    quiet = floor < tol
    assert quiet.mean() > 0.3 and (e_f[quiet] < 50 * tol).all()
    assert (err(g_f[f], g_o[f]) < 50 * tol).all(), err(g_f[f], g_o[f]).max()
    assert (err(g_s[f], g_o[f]) < 50 * tol).all()
    # particles the fast path refused: identical to the Jacobi path by construction
    assert np.array_equal(s_f[~f], s_s[~f]) and np.array_equal(g_f[~f], g_s[~f])


def test_packed_pair_gather_equals_plain_gather():
    """mpm_math.h: p2g_gather_grad_pk -- the p2g.grad gather on packed fp32 pairs (v_pk_fma_f32 on the device, the -DPLB_PK_GATHER=1
    build) -- produces the same 51 sums as the plain sum-factorised gather: the pairing tables (which accumulators share an
    instruction, which half of which factor pair is broadcast) are index bookkeeping that a typo breaks silently.  fp32 both ways; the
    host build of the plain form does not fuse its multiply-adds, hence one rounding per term of slack."""
    from tests import emul
    rng = np.random.default_rng(5)
    n = 3000
    x = rng.uniform(0.05, 0.95, (n, 3))
    g = rng.standard_normal((n, 27, 4)) * rng.uniform(0.1, 10.0, (n, 1, 1))
    assert emul.gather_pk_check(64, 1.0e-4, x, g) < 2e-5
    # one field at a time: a swapped pair shows up as an O(1) difference, not as round-off
    for a in range(4):
        ga = np.zeros_like(g)
        ga[:, :, a] = g[:, :, a]
        assert emul.gather_pk_check(128, 6.1e-5, x, ga) < 2e-5, a
    # the forward gather of g2p on packed pairs (g2p_particle_pk): v', C' and the clamped x'
    gv = rng.standard_normal((n, 27, 3)) * rng.uniform(0.01, 3.0, (n, 1, 1))
    assert emul.g2p_pk_check(64, 1e-4, x, gv) < 2e-5
    for a in range(3):
        ga = np.zeros_like(gv)
        ga[:, :, a] = gv[:, :, a]
        assert emul.g2p_pk_check(128, 5e-5, x, ga) < 2e-5, a


@pytest.mark.parametrize("use_float", [False, True])
def test_nearly_singular_column_rebuild_is_unchanged(use_float):
    """svd_finish rebuilds the weakest column of U by a cross product when sig_min < 1e-3 (crushed or flattened elements).  Round 6
    rewrote that branch as straight-line selects (the column loop made hipcc index a private array at run time: scratch memory in
    every wave of the scatter kernels); the result must be the old one exactly, whichever column is the weakest, ties included."""
    from tests import emul
    rng = np.random.default_rng(11)
    mats = []
    for k in range(600):
        q1, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        q2, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        sig = rng.uniform(0.3, 1.5, 3)
        kind = k % 6
        if kind < 3:
            sig[kind] = rng.choice([0.0, 1e-9, 1e-5, 5e-4])          # one weak direction, each position in turn
        elif kind == 3:
            sig[:2] = 1e-6                                           # two equally weak directions (a tie: the first one is rebuilt)
        elif kind == 4:
            sig[:] = 0.0                                             # F = 0
        if k % 7 == 0:
            q1[:, 0] *= -1                                           # inverted elements too
        mats.append(q1 @ np.diag(sig) @ q2.T - np.eye(3))
    mats.append(np.diag([-1.0, 0.2, -0.1]))                          # already diagonal: Jacobi is a no-op, column 0 is the weak one
    mats.append(np.diag([0.1, -1.0, -1.0]))
    d, rebuilt = emul.svd_singular_check(np.array(mats), use_float)
    assert rebuilt >= 400 and d == 0.0, (d, rebuilt)
