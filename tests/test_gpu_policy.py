"""-m gpu: policy-in-the-loop differentiation (SolverNN).  A small torch MLP maps the reference's
observation (sub-sampled particle x / v + manipulator poses, plb/engine/nn/mlp.py:63-84) to the action at every env
step; d loss / d policy-parameters from the HIP engine (adjoint pushed back through the observation with
add_frame_grad / add_primitive_grad) against the same closed loop run on one torch autograd graph in the oracle."""
import numpy as np
import pytest
import torch

from tests.util import O, oracle_prims, sparse_target
from tests.gpu_util import relerr
from tests.test_gpu_loss import c_sdf

pytestmark = pytest.mark.gpu


def make_policy(obs_dim, act_dim):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(obs_dim, 8), torch.nn.Tanh(), torch.nn.Linear(8, act_dim)).double()
    with torch.no_grad():
        net[2].weight.mul_(4.0)             # actions of order 1 so that some components hit the clamp
    return net


@pytest.mark.parametrize("dtype,ltol,gtol", [("float64", 1e-10, 1e-6), ("float32", 1e-5, 5e-3)])
def test_policy_gradient_matches_oracle(oracle_c, dtype, ltol, gtol):
    from plasticinelab_amd.engine import taichi_env as te
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.optimizer.solver_nn import SolverNN
    n, H, n_obs, vw = 1500, 2, 50, 0.5

    class Sub(te.Shapes):
        def get(self):
            x, c = super().get()
            k = len(x) // n
            return np.ascontiguousarray(x[::k][:n]), c[::k][:n]

    cfg = load_scene("Move", 1)
    cfg.ENV.loss.target_path = ""
    orig, te.Shapes = te.Shapes, Sub
    try:
        env = te.TaichiEnv(cfg, compute_dtype=dtype)
    finally:
        te.Shapes = orig
    env.initialize()
    tgt = sparse_target("Move3D-v1")
    env.loss.load_target_density(grids=tgt)
    env.loss.set_weights(10, 10, 1, True)
    obs_dim = (n // (n // n_obs)) * 6 + 7 * len(env.primitives)
    policy = make_policy(obs_dim, env.primitives.action_dim)
    solver = SolverNN(env, policy, horizon=H, n_observed_particles=n_obs, velocity_weight=vw)
    assert solver.obs.dim == obs_dim
    state0 = env.get_state()["state"]
    loss, grad = solver.forward(state0)

    # ---- the same closed loop on one autograd graph in the oracle
    prims = oracle_prims(cfg)
    s = cfg.SIMULATOR
    sim = O.SimCfg(n_particles=n, yield_stress=s.yield_stress, E=s.E, nu=s.nu, ground_friction=s.ground_friction)
    sdf = torch.as_tensor(c_sdf(oracle_c, tgt, sim.dx).reshape(-1))
    td = torch.as_tensor(tgt.reshape(-1))
    ref_policy = make_policy(obs_dim, env.primitives.action_dim)
    state, mats, poses = O.init_state(env.init_particles), O.materials(sim), O.init_poses(prims)
    idx = torch.as_tensor(solver.obs.index)
    total = 0.0
    for _ in range(H):
        obs = torch.cat([torch.cat([state[0][idx], state[1][idx] * vw], 1).reshape(-1)] + [torch.cat(po[:2]) for po in poses])
        act = torch.clamp(ref_policy(obs), -1.0, 1.0)
        state, poses = O.env_step(sim, prims, 666.0, state, mats, poses, act)
        l, _ = O.compute_loss(sim, O.LossCfg(soft_contact=True), prims, state[0], poses, td, sdf)
        total = total + l
    total.backward()
    ref = np.concatenate([p.grad.numpy().reshape(-1) for p in ref_policy.parameters()])
    assert abs(loss - float(total)) / abs(float(total)) < ltol
    assert np.abs(ref).max() > 0 and relerr(grad, ref) < gtol
