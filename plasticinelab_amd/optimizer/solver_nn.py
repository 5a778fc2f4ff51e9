"""Policy-in-the-loop differentiation: mirror of /root/reference/plb/optimizer/solver_nn.py:3-76 with the
hand-rolled Taichi MLP (plb/engine/nn/mlp.py) replaced by any ``torch.nn.Module``.

At every env step the policy sees the observation the reference's MLP builds (mlp.py:63-84): position and
``velocity_weight`` x velocity of every ``obs_step``-th particle at the step's first frame, then the 7-number pose of
every manipulator; its output, clamped to [-1, 1] (mlp.py:98), is the step's action.  The gradient of the rollout
loss w.r.t. the policy parameters flows

    loss -> (HIP adjoint of the step) -> d loss / d action[t] -> torch autograd through the policy
         -> d loss / d obs[t] -> back into the HIP adjoint of frame t*substeps (particle x, v and manipulator poses)

the last arrow through ``Tape(after_step_grad=...)`` / ``Engine.add_frame_grad`` / ``add_primitive_grad``.
"""
from __future__ import annotations

import numpy as np
import torch

from ..config import CfgNode
from ..engine.taichi_env import TaichiEnv, Tape
from .optim import Adam, Momentum

OPTIMS = {"Adam": Adam, "Momentum": Momentum}


class Observation:
    """mlp.py:31-38,63-84: which particles are observed and how an observation vector is laid out."""

    def __init__(self, env: TaichiEnv, n_observed_particles=200, velocity_weight=1.0):
        self.env = env
        n = env.simulator.n_particles
        self.obs_step = n // n_observed_particles
        self.obs_num = n // self.obs_step
        self.index = np.arange(self.obs_num) * self.obs_step
        self.velocity_weight = float(velocity_weight)
        self.dim = self.obs_num * 6 + 7 * len(env.primitives)

    def read(self, f) -> np.ndarray:
        sim = self.env.simulator
        fr = sim.engine.get_frame(f, want=("x", "v"))
        part = np.concatenate([fr["x"][self.index], fr["v"][self.index] * self.velocity_weight], axis=1).reshape(-1)
        prim = [p.get_state(f)[:7] for p in self.env.primitives]
        return np.concatenate([part] + prim) if prim else part

    def push_grad(self, f, g: np.ndarray):
        """d loss / d obs -> adjoint of frame ``f`` (which must be the resident adjoint frame)."""
        sim = self.env.simulator
        n = sim.n_particles
        gp = g[:self.obs_num * 6].reshape(self.obs_num, 6)
        xa, va = np.zeros((n, 3)), np.zeros((n, 3))
        xa[self.index] = gp[:, :3]
        va[self.index] = gp[:, 3:] * self.velocity_weight
        sim.engine.add_frame_grad(f, xa=xa, va=va)
        base = self.obs_num * 6
        for i in range(len(self.env.primitives)):
            sim.engine.add_primitive_grad(i, f, g[base + 7 * i: base + 7 * i + 7])


class SolverNN:
    def __init__(self, env: TaichiEnv, policy: torch.nn.Module, logger=None, cfg=None, n_observed_particles=200,
                 velocity_weight=1.0, **kwargs):
        self.cfg = self.default_config()
        if cfg is not None:
            self.cfg.merge(dict(cfg), strict=False)
        for k, v in kwargs.items():
            node = self.cfg
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v
        self.cfg.optim.lr *= 0.001                         # solver_nn.py:6-7
        self.cfg.optim.bounds = (-np.inf, np.inf)
        self.env = env
        self.policy = policy.double()
        self.logger = logger
        self.obs = Observation(env, n_observed_particles, velocity_weight)
        self.total_steps = 0

    @classmethod
    def default_config(cls):                               # solver_nn.py:65-76
        return CfgNode({"optim": {"lr": 0.1, "bounds": (-1.0, 1.0), "type": "Adam"}, "n_iters": 100, "softness": 666.0,
                        "horizon": 50, "init_range": 0.0, "init_sampler": "uniform"})

    # ---- flat parameter vector, as nn.get_params / set_params / get_grad (mlp.py:154-183)
    def get_params(self):
        return np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in self.policy.parameters()])

    def set_params(self, flat):
        flat = np.asarray(flat, np.float64)
        o = 0
        with torch.no_grad():
            for p in self.policy.parameters():
                n = p.numel()
                p.copy_(torch.as_tensor(flat[o:o + n].reshape(p.shape), dtype=p.dtype))
                o += n

    def forward(self, sim_state, params=None):             # solver_nn.py:28-43
        env, sim = self.env, self.env.simulator
        if params is not None:
            self.set_params(params)
        env.set_state(sim_state, self.cfg.softness, False)
        if self.logger is not None:
            self.logger.reset()
        for p in self.policy.parameters():
            p.grad = None
        graph = {}                                         # step -> (obs tensor, action tensor)
        A = env.primitives.action_dim

        def after_step_grad(step, first_frame):
            obs_t, act_t = graph.pop(step)
            g = torch.as_tensor(sim.engine.get_action_grad(step + 1)[step].reshape(-1)[:A].copy())
            act_t.backward(g)                              # accumulates into policy.parameters().grad, fills obs_t.grad
            self.obs.push_grad(first_frame, obs_t.grad.numpy())

        with Tape(env, after_step_grad=after_step_grad):
            for i in range(self.cfg.horizon):
                obs_t = torch.tensor(self.obs.read(sim.cur), dtype=torch.float64, requires_grad=True)
                act_t = torch.clamp(self.policy(obs_t), -1.0, 1.0)          # mlp.py:98
                graph[i] = (obs_t, act_t)
                env.step(act_t.detach().numpy())
                self.total_steps += 1
                info = env.compute_loss()
                if self.logger is not None:
                    self.logger.step(None, None, info["reward"], None, i == self.cfg.horizon - 1, info)
        grad = np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).numpy().reshape(-1)
                               for p in self.policy.parameters()])
        return env.loss.loss, grad

    def solve(self, callbacks=()):                         # solver_nn.py:14-63
        env = self.env
        params = self.get_params()
        optim = OPTIMS[self.cfg.optim.type](params, self.cfg.optim)
        env_state = env.get_state()
        self.total_steps = 0
        best, best_loss = None, 1e10
        for _ in range(self.cfg.n_iters):
            self.params = params
            loss, grad = self.forward(env_state["state"], params)
            if loss < best_loss:
                best_loss, best = loss, params.copy()
            params = optim.step(grad)
            for cb in callbacks:
                cb(self, optim, loss, grad)
        env.set_state(**env_state)
        self.set_params(best)
        return best
