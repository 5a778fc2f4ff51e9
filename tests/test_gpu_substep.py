"""-m gpu parity tests of the HIP substep (forward + hand-derived adjoint) against the CPU oracle,
through the C ABI.  Tolerances: float64 engine 1e-9 relative (max-norm); float32 engine 2e-5."""
import numpy as np
import pytest
import torch

from tests.util import O, oracle_scene
from tests.gpu_util import engine_for, load_state, preroll, relerr

pytestmark = pytest.mark.gpu

TOL = {"float64": 1e-9, "float32": 2e-5}


@pytest.fixture(scope="module")
def rolled():
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=2000)
    acts = np.zeros((2, 6)); acts[:, 0] = 0.9; acts[:, 3] = -0.9; acts[:, 1] = 0.3
    state, mats, poses = preroll(sim, prims, x0, acts)
    return sim, prims, state, mats, poses, acts


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_substep_forward_and_adjoint(rolled, dtype):
    sim, prims, state, mats, poses, acts = rolled
    soft = 666.0
    eng = engine_for(sim, prims, dtype=dtype)
    eng.set_softness(soft)
    load_state(eng, 0, state, mats, poses)
    eng.set_action(0, sim.substeps, acts[0])
    eng.substep(0)
    got = eng.get_frame(1)

    # oracle: same substep, with autograd
    a0 = torch.as_tensor(acts[0], dtype=O.DT)
    vel = [O.set_velocity(p, a0[3 * k:3 * k + 3], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, pos, rot, v, w) for p, (pos, rot), (v, w) in zip(prims, poses, vel)]
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
    nin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in nxt]
    out = O.substep(sim, prims, soft, sin, mats, pin, nin)
    tol = TOL[dtype]
    for key, ref in zip(("x", "v", "C", "F"), out):
        assert relerr(got[key], ref.detach().numpy()) < tol, key
    # primitive pose at frame 1
    for k in range(len(prims)):
        st = eng.get_primitive_state(k, 1)
        assert np.allclose(st[:3], nxt[k][0].numpy(), atol=1e-14) and np.allclose(st[3:7], nxt[k][1].numpy(), atol=1e-14)

    g = torch.Generator().manual_seed(1)
    cot = [torch.randn(t.shape, generator=g, dtype=O.DT) for t in out]
    obj = sum((o * c).sum() for o, c in zip(out, cot))
    inputs = list(sin) + [t for pr in pin for t in pr] + [t for pr in nin for t in pr]
    gs = torch.autograd.grad(obj, inputs, allow_unused=True)
    gs = [torch.zeros_like(t) if gg is None else gg for gg, t in zip(gs, inputs)]

    eng.grad_begin(1)
    eng.add_frame_grad(1, xa=cot[0].numpy(), va=cot[1].numpy(), Ca=cot[2].numpy(), Fa=cot[3].numpy())
    eng.substep_grad(0)
    ga = eng.get_frame_grad(0)
    for key, ref in zip(("x", "v", "C", "F"), gs[:4]):
        assert relerr(ga[key], ref.numpy()) < 5 * tol, key
    P = len(prims)
    for k in range(P):
        g0, g1 = eng.get_primitive_grad(k, 0)[:7], eng.get_primitive_grad(k, 1)[:7]
        ref0 = np.concatenate([gs[4 + 2 * k].numpy(), gs[5 + 2 * k].numpy()])
        ref1 = np.concatenate([gs[4 + 2 * P + 2 * k].numpy(), gs[5 + 2 * P + 2 * k].numpy()])
        assert relerr(g0, ref0) < 5 * tol and relerr(g1, ref1) < 5 * tol
    eng.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_mass_and_momentum_conservation(rolled, dtype):
    """Known-answer: sum(grid_m) after the mass scatter equals N * p_mass (SURVEY 8c KAT 1)."""
    sim, prims, state, mats, poses, acts = rolled
    eng = engine_for(sim, prims, dtype=dtype)
    load_state(eng, 0, state, mats, poses)
    gm = eng.grid_mass(0)
    assert abs(gm.sum() / (sim.n_particles * sim.p_mass) - 1) < (1e-12 if dtype == "float64" else 1e-5)
    ref = O.compute_grid_m(sim, state[0]).reshape(sim.n_grid, sim.n_grid, sim.n_grid).numpy()
    assert relerr(gm, ref) < TOL[dtype]
    nodes, blocks = eng.grid_stats(0)
    assert nodes == int((ref > 0).sum())
    eng.close()
