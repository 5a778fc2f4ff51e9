"""CPU tier: the oracle against Taichi-generated vectors, WHEN THEY EXIST.

tests/golden/make_taichi_golden.py produces tests/golden/taichi_{move_v1,substep,semantics}.npz where Taichi 0.7.14
and the reference are installed (not here: no network, no wheel).  Until those files are committed every test below
skips with "parity unpinned" -- which is the status DESIGN.md reports.  With them, this file is what turns the
oracle's three unverified Taichi-semantics assumptions and the hot path's parity into checked facts."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, O, oracle_scene, sparse_target

UNPINNED = "parity unpinned: tests/golden/{} not generated (run tests/golden/make_taichi_golden.py where Taichi 0.7.14 is installed)"


def load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(UNPINNED.format(name))
    return np.load(path, allow_pickle=False)


def test_generator_is_shipped_and_self_describing():
    src = open(os.path.join(GOLDEN, "make_taichi_golden.py")).read()
    assert "taichi==0.7.14" in src and "ti.Tape" in src and "substep_grad" in src
    compile(src, "make_taichi_golden.py", "exec")               # at least it parses without Taichi


def test_autodiff_semantics_match_taichi():
    g = load("taichi_semantics.npz")
    # max / min on an exact tie: the oracle's ti_max / ti_min route the adjoint to the SECOND argument
    for name, fn in (("max", O.ti_max), ("min", O.ti_min)):
        a = torch.tensor(0.25, dtype=O.DT, requires_grad=True)
        b = torch.tensor(0.25, dtype=O.DT, requires_grad=True)
        fn(a, b).backward()
        assert np.allclose([float(a.grad), float(b.grad)], g[f"{name}_tie_grad"]), name
    # atomic_min differentiated as an add: every contributing value gets the adjoint
    v = torch.tensor([0.7, 0.2, 0.9, 0.2], dtype=O.DT, requires_grad=True)
    (O.AtomicMinAsAdd.apply(100000.0, v) ** 2).backward()
    assert np.allclose(v.grad.numpy(), g["atomic_min_grad"])


def test_substep_and_adjoint_match_taichi():
    g = load("taichi_substep.npz")
    cfg, sim, prims, _ = oracle_scene("Move", 1)
    T = lambda a: torch.tensor(np.asarray(a), dtype=O.DT, requires_grad=True)
    state = (T(g["x"]), T(g["v"]), T(g["C"]), T(g["F"]))
    act = torch.tensor(g["action"], dtype=O.DT)
    poses = [(T(p[:3]), torch.tensor(p[3:7], dtype=O.DT)) for p in g["prim"]]
    vel = [O.set_velocity(p, act[3 * k:3 * k + 3], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, po[0], po[1], vw[0], vw[1]) for p, po, vw in zip(prims, poses, vel)]
    out = O.substep(sim, prims, 666.0, state, O.materials(sim), poses, nxt)
    for a, name in zip(out, ("x1", "v1", "C1", "F1")):
        ref = g[name]
        assert np.abs(a.detach().numpy() - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30), name
    obj = sum((o * torch.tensor(g[c])).sum() for o, c in zip(out, ("cot_x", "cot_v", "cot_C", "cot_F")))
    grads = torch.autograd.grad(obj, list(state))
    for a, name in zip(grads, ("xa", "va", "Ca", "Fa")):
        ref = g[name]
        assert np.abs(a.numpy() - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-30), name


def test_move_v1_rollout_matches_taichi():
    g = load("taichi_move_v1.npz")
    ours = np.load(os.path.join(GOLDEN, "rollout_move_v1.npz"))           # the oracle's rollout of the same actions
    assert np.array_equal(ours["actions"], g["actions"])
    assert abs(float(ours["loss"]) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert np.abs(ours["grad"] - g["grad"]).max() <= 1e-4 * np.abs(g["grad"]).max()          # north star: 1e-4
    assert np.abs(ours["x_final"] - g["x_final"]).max() <= 1e-6


def test_contact_loss_gradients_match_taichi():
    g = load("taichi_semantics.npz")
    cfg, sim, prims, x0 = oracle_scene("Move", 1)
    zeros = torch.zeros(sim.n_grid ** 3, dtype=O.DT)
    for soft in (False, True):
        L, grad, *_ = O.rollout_loss_and_grad(sim, O.LossCfg(sdf_weight=0.0, density_weight=0.0, contact_weight=1.0, soft_contact=soft), prims, 666.0,
                                              O.init_state(x0), O.materials(sim), O.init_poses(prims),
                                              torch.as_tensor(g["contact_actions"], dtype=O.DT), zeros, zeros)
        tag = "soft" if soft else "hard"
        assert abs(L - float(g[f"contact_{tag}_loss"])) <= 1e-8 * max(abs(float(g[f"contact_{tag}_loss"])), 1e-30)
        assert np.abs(grad.numpy() - g[f"contact_{tag}_grad"]).max() <= 1e-6 * max(np.abs(g[f"contact_{tag}_grad"]).max(), 1e-30)
