"""Gradient-based action-sequence solver; mirror of /root/reference/plb/optimizer/solver.py:13-101.
``Solver.forward`` is the caller whose ``(loss, grad)`` outputs define parity for this path."""
from __future__ import annotations

import numpy as np

from ..config import CfgNode
from ..engine.taichi_env import TaichiEnv, Tape
from .optim import Adam, Momentum

OPTIMS = {"Adam": Adam, "Momentum": Momentum}


class Solver:
    def __init__(self, env: TaichiEnv, logger=None, cfg=None, **kwargs):
        self.cfg = self.default_config()
        if cfg is not None:
            self.cfg.merge(dict(cfg), strict=False)
        for k, v in kwargs.items():                      # dotted keys like "optim.lr" as in the reference
            node = self.cfg
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v
        self.optim_cfg = self.cfg.optim
        self.env = env
        self.logger = logger
        self.total_steps = 0

    @classmethod
    def default_config(cls):                             # solver.py:74-83
        return CfgNode({"optim": {"lr": 0.1, "bounds": (-1.0, 1.0), "type": "Adam"},
                        "n_iters": 100, "softness": 666.0, "horizon": 50, "init_range": 0.0,
                        "init_sampler": "uniform",
                        "checkpoint_segment": 0})      # > 0: segment-checkpointed backward (optimizer/checkpoint.py)

    def forward(self, sim_state, action):                # solver.py:31-44
        env = self.env
        if self.logger is not None:
            self.logger.reset()
        if self.cfg.checkpoint_segment and self.cfg.checkpoint_segment < len(action):
            from .checkpoint import forward_checkpointed
            self.total_steps += len(action)
            return forward_checkpointed(env, sim_state, action, self.cfg.checkpoint_segment, self.cfg.softness)
        env.set_state(sim_state, self.cfg.softness, False)
        with Tape(env):
            for i in range(len(action)):
                env.step(action[i])
                self.total_steps += 1
                loss_info = env.compute_loss()
                if self.logger is not None:
                    self.logger.step(None, None, loss_info["reward"], None, i == len(action) - 1, loss_info)
        return env.loss.loss, env.primitives.get_grad(len(action))

    def solve(self, init_actions=None, callbacks=()):    # solver.py:21-61
        env = self.env
        if init_actions is None:
            init_actions = self.init_actions(env, self.cfg)
        optim = OPTIMS[self.optim_cfg.type](init_actions, self.optim_cfg)
        env_state = env.get_state()
        self.total_steps = 0
        best_action, best_loss = None, 1e10
        actions = init_actions
        for _ in range(self.cfg.n_iters):
            self.params = actions.copy()
            loss, grad = self.forward(env_state["state"], actions)
            if loss < best_loss:
                best_loss, best_action = loss, actions.copy()
            actions = optim.step(grad)
            for cb in callbacks:
                cb(self, optim, loss, grad)
        env.set_state(**env_state)
        return best_action

    @staticmethod
    def init_actions(env, cfg):                          # solver.py:63-71
        if cfg.init_sampler != "uniform":
            raise NotImplementedError
        return np.random.uniform(-cfg.init_range, cfg.init_range, size=(cfg.horizon, env.primitives.action_dim))


def solve_action(env, path=None, logger=None, args=None, n_iters=None, lr=0.1, optim="Adam", softness=666.0):
    """solver.py:86-101 without the rendering tail (renderer is out of scope)."""
    env.reset()
    taichi_env = env.unwrapped.taichi_env
    T = env._max_episode_steps
    if args is not None:
        n_iters = (args.num_steps + T - 1) // T
        lr, optim, softness = args.lr, args.optim, args.softness
    solver = Solver(taichi_env, logger, None, n_iters=n_iters or 1, softness=softness, horizon=T,
                    **{"optim.lr": lr, "optim.type": optim, "init_range": 0.0001})
    return solver.solve()
