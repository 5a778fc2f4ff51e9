"""Host-side mirror of plb.engine.primitive (Primitive / Primitives).

Same names, argument meaning and error behaviour as the reference
(/root/reference/plb/engine/primitive/primive_base.py:9-224 and
primitives.py:262-320); the pose trajectories, action buffers and their
adjoints live on the GPU inside the engine, this class only configures and
reads them through the C ABI.
"""
from __future__ import annotations

from typing import Iterable, List

import numpy as np

from ..config import CfgNode, as_value

# per-shape geometric parameters and their defaults (primitives.py default_config of each class)
_SHAPE_PARAMS = {
    "Sphere": (("radius",), (1.0,)),
    "Capsule": (("h", "r"), (0.06, 0.03)),
    "RollingPin": (("h", "r"), (0.06, 0.03)),        # a Capsule with its own forward_kinematics (primitives.py:64-80)
    "Cylinder": (("h", "r"), (0.2, 0.1)),
    "Torus": (("tx", "ty"), (0.2, 0.1)),
    "Box": (("size",), ((0.1, 0.1, 0.1),)),
    # two Capsules `gap` apart with a 7-dim action (primitives.py:83-154); the gap itself is part of the state
    "Chopsticks": (("h", "r", "minimal_gap"), (0.06, 0.03, 0.06)),
}
_UNSUPPORTED = {}


def _default_cfg(shape: str) -> CfgNode:
    """Primitive.default_config (primive_base.py:209-224) + the shape's extras."""
    cfg = CfgNode({
        "shape": shape, "init_pos": (0.3, 0.3, 0.3), "init_rot": (1.0, 0.0, 0.0, 0.0),
        "color": (0.3, 0.3, 0.3), "lower_bound": (0.0, 0.0, 0.0), "upper_bound": (1.0, 1.0, 1.0),
        "friction": 0.9, "variations": None, "action": {"dim": 0, "scale": ()},
    })
    names, defaults = _SHAPE_PARAMS[shape]
    for n, d in zip(names, defaults):
        cfg[n] = d
    if shape == "Chopsticks":
        cfg["init_gap"] = 0.06                      # primitives.py:152
    return cfg


class _Field0:
    """Read-only stand-in for a 0-d Taichi field: ``p.min_dist[None]`` as in the reference (loss.py:123-135)."""

    def __init__(self, read):
        self._read = read

    def __getitem__(self, key):
        if key is not None and key != ():
            raise IndexError("0-d field: index with [None]")
        return float(self._read())

    def __float__(self):
        return float(self._read())

    def to_numpy(self):
        return np.asarray(float(self._read()))


class Primitive:
    """One rigid manipulator (reference class Primitive and its shape subclasses)."""

    state_dim = 7

    def __init__(self, cfg, index: int, max_timesteps: int = 1024):
        shape = cfg["shape"]
        if shape in _UNSUPPORTED:
            raise NotImplementedError(f"primitive shape {shape} is not built yet: {_UNSUPPORTED[shape]}")
        if shape not in _SHAPE_PARAMS:
            raise NotImplementedError(f"unknown primitive shape {shape!r}")
        self.cfg = _default_cfg(shape)
        self.cfg.merge({k: (dict(v) if isinstance(v, dict) else v) for k, v in dict(cfg).items()}, strict=True)
        for k in ("init_pos", "init_rot", "lower_bound", "upper_bound", "color"):
            self.cfg[k] = tuple(as_value(self.cfg[k]))
        self.cfg.action["scale"] = tuple(as_value(self.cfg.action.scale))
        self.shape = shape
        self.index = index
        self.dim = 3
        self.max_timesteps = max_timesteps
        self.action_dim = int(self.cfg.action.dim)
        if shape == "Chopsticks":                   # primitives.py:84,92
            self.state_dim = 8
            assert self.action_dim == 7, "Chopsticks: 3 linear, 3 angle, 1 for grasp"
        self._softness = 0.0
        self._engine = None

    # ---- description handed to the engine
    def params(self):
        names, _ = _SHAPE_PARAMS[self.shape]
        vals: List[float] = []
        for n in names:
            v = as_value(self.cfg[n])
            vals.extend(v if isinstance(v, (tuple, list)) else [v])
        return tuple(float(v) for v in vals)

    def describe(self) -> dict:
        return dict(shape=self.shape, action_dim=self.action_dim, params=self.params(),
                    friction=float(self.cfg.friction), action_scale=tuple(float(s) for s in self.cfg.action.scale),
                    lower_bound=tuple(float(v) for v in self.cfg.lower_bound),
                    upper_bound=tuple(float(v) for v in self.cfg.upper_bound))

    def _bind(self, engine):
        self._engine = engine

    def _eng(self):
        if self._engine is None:
            raise RuntimeError("primitive is not attached to a simulator yet (construct MPMSimulator first)")
        return self._engine

    # ---- reference API
    @property
    def init_state(self):                       # primive_base.py:153-155; Chopsticks primitives.py:131-133
        st = tuple(self.cfg.init_pos) + tuple(self.cfg.init_rot)
        if self.shape == "Chopsticks":
            st = st + (float(self.cfg.init_gap),)
        return st

    def initialize(self):                       # primive_base.py:157-164
        self.set_state(0, self.init_state)

    def get_state(self, f):                     # primive_base.py:143-146; Chopsticks appends the gap (:135-136)
        return self._eng().get_primitive_state(self.index, f)[:self.state_dim]

    def set_state(self, f, state):              # primive_base.py:148-151; Chopsticks :143-146
        state = np.asarray(state, np.float64).reshape(-1)
        if self.shape == "Chopsticks":
            assert len(state) == 8
        ss = self._eng().get_primitive_state(self.index, f)
        ss[:len(state)] = state
        self._eng().set_primitive_state(self.index, f, ss)

    @property
    def friction(self):
        return float(self.cfg.friction)

    @property
    def softness(self):
        return self._softness

    def set_velocity(self, s, n_substeps):      # primive_base.py:184-192: v, w of env step s from action_buffer[s]
        if self.action_dim > 0:
            self._eng().set_velocity(self.index, s, n_substeps)

    def sdf(self, f, grid_pos):                 # primive_base.py:57-60 (+ the shape's own sdf, primitives.py)
        """Signed distance of ``grid_pos`` ((3,) or (n,3)) to the primitive at its pose of frame ``f``."""
        pts = np.asarray(grid_pos, np.float64)
        d = self._eng().primitive_sdf(self.index, f, pts.reshape(-1, 3))
        return float(d[0]) if pts.ndim == 1 else d

    @property
    def min_dist(self):                         # primive_base.py:37 / loss.py:123-135: after the last compute_loss
        return _Field0(lambda: self._eng().loss_contact_scalars()[0][self.index])

    @property
    def dist_norm(self):                        # primive_base.py:39 / loss.py:116-121 (soft contact loss)
        return _Field0(lambda: self._eng().loss_contact_scalars()[1][self.index])

    def get_action_grad(self, s, n):            # primive_base.py:200-206
        if self.action_dim == 0:
            return None
        g = self._eng().get_action_grad(s + n)
        ofs = sum(self._eng().action_dims[:self.index])
        return g[s:s + n, ofs:ofs + self.action_dim]


class Primitives:
    """reference class Primitives (primitives.py:262-320)."""

    def __init__(self, cfgs: Iterable, max_timesteps: int = 1024):
        self.primitives = [Primitive(c, i, max_timesteps) for i, c in enumerate(cfgs)]
        self.action_dims = [0]
        for p in self.primitives:
            self.action_dims.append(self.action_dims[-1] + p.action_dim)
        self.n = len(self.primitives)
        self._engine = None
        self._softness = 0.0

    def _bind(self, engine):
        self._engine = engine
        for p in self.primitives:
            p._bind(engine)

    @property
    def action_dim(self):
        return self.action_dims[-1]

    @property
    def state_dim(self):
        return sum(p.state_dim for p in self.primitives)

    def set_action(self, s, n_substeps, action):            # primitives.py:289-293
        action = np.asarray(action, np.float64).reshape(-1).clip(-1, 1)
        assert len(action) == self.action_dims[-1]
        self._engine.set_action(s, n_substeps, action)

    def get_grad(self, n):                                  # primitives.py:295-301
        return self._engine.get_action_grad(n)

    def set_softness(self, softness=666.0):                 # primitives.py:303-305
        self._softness = float(softness)
        for p in self.primitives:
            p._softness = self._softness
        if self._engine is not None:
            self._engine.set_softness(self._softness)

    def get_softness(self):                                 # primitives.py:307-308
        return self._softness

    def __getitem__(self, item):
        if isinstance(item, tuple):
            item = item[0]
        return self.primitives[item]

    def __len__(self):
        return len(self.primitives)

    def __iter__(self):
        return iter(self.primitives)

    def initialize(self):                                   # primitives.py:318-320
        for p in self.primitives:
            p.initialize()
        self.set_softness(self._softness)
