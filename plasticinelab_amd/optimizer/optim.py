"""First-order optimisers over a flat or (horizon, action_dim) numpy parameter array.

Public surface as in the reference (/root/reference/plb/optimizer/optim.py:5-77): ``Adam(parameters, cfg, **kw)`` /
``Momentum(...)``, config keys ``lr, bounds, type`` (+ ``momentum`` / ``beta_1, beta_2, epsilon``), and
``step(grads) -> new parameters`` which updates ``parameters`` in place and clips to ``bounds``.  The implementation
is organised around one hook, ``direction(g, t)``: the (bias-corrected) descent direction for gradient ``g`` at
update count ``t``; running statistics live in ``self.state``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from ..config import CfgNode

_BASE_KEYS = {"lr": 0.1, "bounds": (-1.0, 1.0), "type": ""}


class Optimizer:
    extra_keys: Dict[str, float] = {}

    def __init__(self, parameters: np.ndarray, cfg=None, **kwargs):
        self.cfg = CfgNode(dict(self.default_config()))
        for source in (cfg, kwargs):
            if source:
                self.cfg.merge(dict(source), strict=False)
        self.parameters = parameters
        self.state: Dict[str, np.ndarray] = {}
        self.updates = 0

    @classmethod
    def default_config(cls) -> dict:
        return {**_BASE_KEYS, **cls.extra_keys}

    # the two knobs the solvers read back -- and may set (an lr schedule writes ``optim.lr = ...`` in the
    # reference, where they are plain attributes: optim.py:12-13); both write through to ``cfg``
    @property
    def lr(self):
        return self.cfg.lr

    @lr.setter
    def lr(self, value):
        self.cfg.lr = float(value)

    @property
    def bounds(self):
        return tuple(self.cfg.bounds)

    @bounds.setter
    def bounds(self, value):
        lo, hi = value
        self.cfg.bounds = (float(lo), float(hi))

    @property
    def iter(self):                                  # reference name of the update counter (optim.py:36,60)
        return self.updates

    @iter.setter
    def iter(self, value):
        self.updates = int(value)

    def _stat(self, name: str) -> np.ndarray:
        """Running statistic ``name`` (float64, shaped like the parameters, zero at first use)."""
        if name not in self.state:
            self.state[name] = np.zeros(self.parameters.shape, np.float64)
        return self.state[name]

    def direction(self, g: np.ndarray, t: int) -> np.ndarray:
        raise NotImplementedError

    def step(self, grads) -> np.ndarray:
        g = np.asarray(grads, np.float64)
        if g.shape != self.parameters.shape:
            raise AssertionError(f"gradient shape {g.shape} != parameter shape {self.parameters.shape}")
        lo, hi = self.bounds
        np.clip(self.parameters - self.lr * self.direction(g, self.updates), lo, hi, out=self.parameters)
        self.updates += 1
        return np.array(self.parameters)


class Momentum(Optimizer):
    """Exponential moving average of the gradient (weight ``momentum`` on the past)."""
    extra_keys = {"momentum": 0.9}

    @property
    def momentum_buffer(self):                       # reference attribute (optim.py:37)
        return self._stat("avg")

    def direction(self, g, t):
        avg = self._stat("avg")
        avg *= self.cfg.momentum
        avg += (1.0 - self.cfg.momentum) * g
        return avg


class Adam(Optimizer):
    """Adam with bias correction; ``epsilon`` is added to the root of the corrected second moment."""
    extra_keys = {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-8}

    @property
    def momentum_buffer(self):                       # reference attributes (optim.py:61-62)
        return self._stat("first")

    @property
    def v_buffer(self):
        return self._stat("second")

    def direction(self, g, t):
        first, second = self._stat("first"), self._stat("second")
        for stat, beta, sample in ((first, self.cfg.beta_1, g), (second, self.cfg.beta_2, g * g)):
            stat *= beta
            stat += (1.0 - beta) * sample
        unbias = [1.0 - beta ** (t + 1) for beta in (self.cfg.beta_1, self.cfg.beta_2)]
        return (first / unbias[0]) / (np.sqrt(second / unbias[1]) + self.cfg.epsilon)
