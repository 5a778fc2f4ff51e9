"""-m gpu: movable Capsule (Writer tasks) and Torus (Torus tasks) manipulators, tilted and driven with 6-dof
actions, through the whole Solver.forward path -- loss, d loss/d actions incl. the rotation components and the
soft contact-loss adjoint -- against the oracle run live (float64 engine, 1500 particles, 2 env steps)."""
import numpy as np
import pytest
import torch

from tests.util import O, oracle_prims, sparse_target
from tests.gpu_util import relerr
from tests.test_gpu_loss import c_sdf

pytestmark = pytest.mark.gpu


def scene(shape, kw):
    from plasticinelab_amd.envs.scenes import load_scene
    cfg = load_scene("Move", 1)
    rot = np.array([0.9, 0.2, -0.3, 0.25]); rot /= np.linalg.norm(rot)
    prims = []
    for p in cfg.PRIMITIVES:
        d = {"shape": shape, "init_pos": tuple(p["init_pos"]), "init_rot": tuple(float(r) for r in rot), "friction": 0.9,
             "action": {"dim": 6, "scale": (0.01,) * 6}}
        d.update(kw)
        prims.append(d)
    cfg["PRIMITIVES"] = prims
    cfg.ENV.loss.target_path = ""
    return cfg


@pytest.mark.parametrize("shape,kw", [("Capsule", dict(h=0.06, r=0.03)), ("Torus", dict(tx=0.05, ty=0.02))])
@pytest.mark.parametrize("soft", [False, True])
def test_movable_shape_rollout(oracle_c, shape, kw, soft):
    from plasticinelab_amd.engine import taichi_env as te
    from plasticinelab_amd.optimizer.solver import Solver
    n = 1500

    class Sub(te.Shapes):
        def get(self):
            x, c = super().get()
            k = len(x) // n
            return np.ascontiguousarray(x[::k][:n]), c[::k][:n]

    cfg = scene(shape, kw)
    orig, te.Shapes = te.Shapes, Sub
    try:
        env = te.TaichiEnv(cfg, compute_dtype="float64")
    finally:
        te.Shapes = orig
    env.initialize()
    tgt = sparse_target("Move3D-v1")
    env.loss.load_target_density(grids=tgt)
    env.loss.set_weights(10, 10, 1, soft)
    acts = np.array([[0.9, 0.3, 0.1, 0.5, -0.4, 0.3, -0.9, 0.2, -0.1, -0.3, 0.6, 0.2],
                     [0.5, -0.2, 0.3, -0.6, 0.2, 0.4, -0.4, 0.1, 0.2, 0.3, -0.5, -0.2]])
    state0 = env.get_state()["state"]
    loss, grad = Solver(env, None, None, softness=666.0, horizon=2).forward(state0, acts)

    prims = oracle_prims(cfg)
    s = cfg.SIMULATOR
    sim = O.SimCfg(n_particles=n, yield_stress=s.yield_stress, E=s.E, nu=s.nu, ground_friction=s.ground_friction)
    sdf = c_sdf(oracle_c, tgt, sim.dx)
    L, g, *_ = O.rollout_loss_and_grad(sim, O.LossCfg(soft_contact=soft), prims, 666.0, O.init_state(env.init_particles),
                                       O.materials(sim), O.init_poses(prims), torch.as_tensor(acts, dtype=O.DT),
                                       torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1)))
    assert abs(loss - L) / abs(L) < 1e-10
    g = g.numpy()
    assert np.abs(g[:, 3:6]).max() > 0                       # rotation actions do carry gradient
    assert relerr(grad, g) < 1e-7


def test_rollingpin_rollout(oracle_c):
    """Rollingpin-v1 scene (subsampled): RollingPin kinematics + Capsule contact, loss and action gradient."""
    from plasticinelab_amd.engine import taichi_env as te
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.optimizer.solver import Solver
    n = 1500

    class Sub(te.Shapes):
        def get(self):
            x, c = super().get()
            k = len(x) // n
            return np.ascontiguousarray(x[::k][:n]), c[::k][:n]

    cfg = load_scene("Rollingpin", 1)
    cfg.ENV.loss.target_path = ""
    orig, te.Shapes = te.Shapes, Sub
    try:
        env = te.TaichiEnv(cfg, compute_dtype="float64")
    finally:
        te.Shapes = orig
    env.initialize()
    tgt = sparse_target("Move3D-v1")              # any 64^3 target exercises the same code
    env.loss.load_target_density(grids=tgt)
    env.loss.set_weights(10, 10, 1, True)
    acts = np.array([[0.8, -0.5, -0.6], [0.6, 0.4, -0.3]])
    state0 = env.get_state()["state"]
    loss, grad = Solver(env, None, None, softness=666.0, horizon=2).forward(state0, acts)
    prims = oracle_prims(cfg)
    s = cfg.SIMULATOR
    sim = O.SimCfg(n_particles=n, yield_stress=s.yield_stress, E=s.E, nu=s.nu, ground_friction=s.ground_friction)
    sdf = c_sdf(oracle_c, tgt, sim.dx)
    L, g, *_ = O.rollout_loss_and_grad(sim, O.LossCfg(soft_contact=True), prims, 666.0, O.init_state(env.init_particles),
                                       O.materials(sim), O.init_poses(prims), torch.as_tensor(acts, dtype=O.DT),
                                       torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1)))
    assert abs(loss - L) / abs(L) < 1e-10
    assert np.abs(g.numpy()).max() > 0 and relerr(grad, g.numpy()) < 1e-7


@pytest.mark.parametrize("dtype,ltol,gtol", [("float64", 1e-10, 1e-7), ("float32", 1e-5, 2e-3)])
def test_chopsticks_rollout(oracle_c, dtype, ltol, gtol):
    """Chopsticks-v1 scene (subsampled): double-capsule contact, body-frame rotation, gap degree of freedom driven
    by the 7th action component (opened in step 1 so the minimal-gap clamp is inactive, closed in steps 2-3)."""
    from plasticinelab_amd.engine import taichi_env as te
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.optimizer.solver import Solver
    n = 1500

    class Sub(te.Shapes):
        def get(self):
            x, c = super().get()
            k = len(x) // n
            return np.ascontiguousarray(x[::k][:n]), c[::k][:n]

    cfg = load_scene("Chopsticks", 1)
    cfg.ENV.loss.target_path = ""
    orig, te.Shapes = te.Shapes, Sub
    try:
        env = te.TaichiEnv(cfg, compute_dtype=dtype)
    finally:
        te.Shapes = orig
    env.initialize()
    assert env.primitives[0].get_state(0).shape == (8,) and env.primitives[0].get_state(0)[7] == 0.06
    tgt = sparse_target("Move3D-v1")              # any 64^3 target exercises the same code
    env.loss.load_target_density(grids=tgt)
    env.loss.set_weights(10, 10, 1, True)
    acts = np.array([[0.6, -0.2, -0.2, 0.5, -0.4, 0.3, -0.8],
                     [0.5, 0.1, 0.1, -0.3, 0.2, 0.4, 0.3],
                     [-0.4, 0.2, 0.2, 0.2, -0.1, -0.3, 0.3]])
    state0 = env.get_state()["state"]
    loss, grad = Solver(env, None, None, softness=666.0, horizon=3).forward(state0, acts)
    prims = oracle_prims(cfg)
    s = cfg.SIMULATOR
    sim = O.SimCfg(n_particles=n, yield_stress=s.yield_stress, E=s.E, nu=s.nu, ground_friction=s.ground_friction,
                   gravity=tuple(s.gravity))
    sdf = c_sdf(oracle_c, tgt, sim.dx)
    L, g, _, poses, _ = O.rollout_loss_and_grad(sim, O.LossCfg(soft_contact=True), prims, 666.0, O.init_state(env.init_particles),
                                                O.materials(sim), O.init_poses(prims), torch.as_tensor(acts, dtype=O.DT),
                                                torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1)))
    st = env.primitives[0].get_state(3 * sim.substeps)
    ref = np.concatenate([t.numpy().reshape(-1) for t in poses[-1][0]])
    assert np.abs(st - ref).max() < 1e-12 and 0.06 < st[7] < 0.1
    assert abs(loss - L) / abs(L) < ltol
    g = g.numpy()
    assert np.abs(g[:, 6]).min() > 0 and np.abs(g[:, 3:6]).max() > 0      # gap and rotation actions carry gradient
    assert relerr(grad, g) < gtol
    for c in range(7):                                                     # every action component on its own scale
        assert relerr(grad[:, c], g[:, c]) < 30 * gtol
