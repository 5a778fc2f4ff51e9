"""numpy optimisers over an action sequence; same update rules and config keys as
/root/reference/plb/optimizer/optim.py:5-77."""
from __future__ import annotations

import numpy as np

from ..config import CfgNode


class Optimizer:
    defaults = {"lr": 0.1, "bounds": (-1.0, 1.0), "type": ""}

    def __init__(self, parameters: np.ndarray, cfg=None, **kwargs):
        self.cfg = CfgNode(dict(self.default_config()))
        if cfg is not None:
            self.cfg.merge(dict(cfg), strict=False)
        self.cfg.merge(kwargs, strict=False)
        self.lr = self.cfg.lr
        self.bounds = self.cfg.bounds
        self.parameters = parameters
        self.initialize()

    @classmethod
    def default_config(cls):
        out = {}
        for k in reversed(cls.__mro__):
            out.update(getattr(k, "defaults", {}))
        return out

    def initialize(self):
        raise NotImplementedError

    def _step(self, grads):
        raise NotImplementedError

    def step(self, grads):
        assert grads.shape == self.parameters.shape
        self.parameters[:] = self._step(grads).clip(*self.bounds)
        return self.parameters.copy()


class Momentum(Optimizer):
    defaults = {"momentum": 0.9}

    def initialize(self):
        self.momentum_buffer = np.zeros_like(self.parameters, dtype=np.float64)
        self.momentum = self.cfg.momentum

    def _step(self, grads):
        g = self.momentum_buffer * self.momentum + grads * (1 - self.momentum)
        self.momentum_buffer[:] = g
        return self.parameters - self.lr * g


class Adam(Optimizer):
    defaults = {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-8}

    def initialize(self):
        self.momentum_buffer = np.zeros_like(self.parameters, dtype=np.float64)
        self.v_buffer = np.zeros_like(self.momentum_buffer)
        self.iter = 0

    def _step(self, grads):
        b1, b2, eps = self.cfg.beta_1, self.cfg.beta_2, self.cfg.epsilon
        g = grads.reshape(self.parameters.shape)
        m = b1 * self.momentum_buffer + (1 - b1) * g
        v = b2 * self.v_buffer + (1 - b2) * (g * g)
        self.momentum_buffer[:], self.v_buffer[:] = m, v
        m_hat = m / (1 - b1 ** (self.iter + 1))
        v_hat = v / (1 - b2 ** (self.iter + 1))
        self.iter += 1
        return self.parameters - (self.lr * m_hat) / (np.sqrt(v_hat) + eps)
