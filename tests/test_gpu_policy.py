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


# gradient tolerances: measured (round 3) 6e-15 in float64, 8e-7 in float32
@pytest.mark.parametrize("dtype,ltol,gtol", [("float64", 1e-10, 1e-9), ("float32", 1e-5, 1e-4)])
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
    print(f"\n[policy {dtype}] loss rel {abs(loss - float(total)) / abs(float(total)):.2e}  d loss / d parameters max-norm rel {relerr(grad, ref):.2e}")
    assert np.abs(ref).max() > 0 and relerr(grad, ref) < gtol


def test_per_step_autograd_functions():
    """plasticinelab_amd.autograd.RolloutSession: one torch.autograd.Function per env step.  (a) A policy between the
    steps: d loss / d policy-parameters equals SolverNN's (same engine, Tape + hooks); (b) nothing of the adjoint runs
    before backward(); (c) per-step loss weights -- something the Tape form cannot express -- against the weighted sum of
    single-step-loss gradients' defining property: linearity in the weights."""
    from plasticinelab_amd.autograd import RolloutSession
    from plasticinelab_amd.optimizer.solver_nn import SolverNN
    from tests.test_gpu_rollout import make_env_sub
    n_obs, vw, H = 50, 0.5, 3
    env = make_env_sub("Move", 1500, "float64", soft_contact=True)
    obs_dim = (1500 // (1500 // n_obs)) * 6 + 7 * len(env.primitives)
    state0 = env.get_state()["state"]
    solver = SolverNN(env, make_policy(obs_dim, env.primitives.action_dim), horizon=H, n_observed_particles=n_obs, velocity_weight=vw)
    loss_ref, grad_ref = solver.forward(state0)

    policy = make_policy(obs_dim, env.primitives.action_dim)

    def rollout(weights):
        sess = RolloutSession(env, state0, 666.0, n_observed_particles=n_obs, velocity_weight=vw)
        carry = sess.begin()
        losses = []
        for _ in range(H):
            act = torch.clamp(policy(sess.observe(carry)), -1.0, 1.0)
            l, carry = sess.step(act, carry)
            losses.append(l)
        return sess, sum(w * l for w, l in zip(weights, losses)), losses

    def pgrad():
        g = np.concatenate([p.grad.numpy().reshape(-1) for p in policy.parameters()])
        for p in policy.parameters():
            p.grad = None
        return g

    sess, total, losses = rollout([1.0] * H)
    assert abs(float(total) - loss_ref) / abs(loss_ref) < 1e-12
    assert not sess._grad_started                                   # (b) forward only so far
    total.backward()
    g1 = pgrad()
    assert np.abs(grad_ref).max() > 0 and relerr(g1, grad_ref) < 1e-9      # (a)
    with pytest.raises(RuntimeError, match="already been differentiated"):
        sess.step(torch.zeros(env.primitives.action_dim, dtype=torch.float64), torch.zeros((), dtype=torch.float64))
    # (c) weights (w0, w1, w2): gradient is linear in them
    basis = []
    for k in range(H):
        _, t, _ = rollout([1.0 if j == k else 0.0 for j in range(H)])
        t.backward()
        basis.append(pgrad())
    w = [0.3, -1.7, 2.2]
    _, t, _ = rollout(w)
    t.backward()
    gw = pgrad()
    assert relerr(gw, sum(wk * b for wk, b in zip(w, basis))) < 1e-9
    assert relerr(sum(basis), g1) < 1e-9
