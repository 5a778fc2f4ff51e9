"""Mirror of plb.engine.taichi_env.TaichiEnv (/root/reference/plb/engine/taichi_env.py:8-105):
builds primitives, particles, simulator and loss from a config tree and keeps the reference's
copy-mode / tape-mode stepping contract.  The renderer and the Taichi MLP are out of scope
(SURVEY section 2 rows 8-9); ``render`` raises.

Differentiation: the reference wraps a rollout in ``ti.Tape(loss=env.loss.loss)``.  Here the
same call sequence is recorded by ``Tape`` (below) and replayed in reverse through the
hand-written adjoint kernels; ``plasticinelab_amd.autograd.RolloutLoss`` exposes it as a
``torch.autograd.Function``.
"""
from __future__ import annotations

import numpy as np

from .losses import Loss
from .mpm_simulator import MPMSimulator
from .primitives import Primitives
from .shapes import Shapes


class TaichiEnv:
    def __init__(self, cfg, nn=False, loss=True, compute_dtype=None, device=None):
        if nn:
            raise NotImplementedError("the Taichi MLP policy (plb/engine/nn/mlp.py) is out of scope; use a torch policy")
        self.cfg = cfg.ENV
        self.primitives = Primitives(cfg.PRIMITIVES, max_timesteps=int(cfg.SIMULATOR.max_steps))
        self.shapes = Shapes(cfg.SHAPES)
        self.init_particles, self.particle_colors = self.shapes.get()
        cfg.SIMULATOR.n_particles = len(self.init_particles)          # taichi_env.py:28-29
        self.n_particles = len(self.init_particles)
        self.simulator = MPMSimulator(cfg.SIMULATOR, self.primitives, compute_dtype=compute_dtype, device=device)
        self.renderer = None
        self.loss = Loss(cfg.ENV.loss, self.simulator) if loss else None
        self._is_copy = True
        self._tape = None

    def set_copy(self, is_copy: bool):
        self._is_copy = is_copy

    def initialize(self):                                             # taichi_env.py:46-58
        self.primitives.initialize()
        self.simulator.initialize()
        if self.loss:
            self.loss.initialize()
        self.simulator.reset(self.init_particles)
        if self.loss:
            self.loss.clear()

    def render(self, mode="human", **kwargs):
        raise NotImplementedError("rendering is outside the accelerated path (SURVEY section 2, row 8)")

    def step(self, action=None):                                      # taichi_env.py:78-81
        if action is not None:
            action = np.array(action)
        start = 0 if self._is_copy else self.simulator.cur
        self.simulator.step(is_copy=self._is_copy, action=action)
        if self._tape is not None:
            self._tape._record_step(start)

    def compute_loss(self):                                           # taichi_env.py:83-89
        assert self.loss is not None
        if self._is_copy:
            self.loss.clear()
            return self.loss.compute_loss(0)
        info = self.loss.compute_loss(self.simulator.cur)
        if self._tape is not None:
            self._tape._record_loss(self.simulator.cur)
        return info

    def get_state(self):                                              # taichi_env.py:91-97
        assert self.simulator.cur == 0
        return {"state": self.simulator.get_state(0), "softness": self.primitives.get_softness(),
                "is_copy": self._is_copy}

    def set_state(self, state, softness, is_copy):                    # taichi_env.py:99-106
        self.simulator.cur = 0
        self.simulator.set_state(0, state)
        self.primitives.set_softness(softness)
        self._is_copy = is_copy
        if self.loss:
            self.loss.reset()
            self.loss.clear()


class Tape:
    """Stand-in for ``ti.Tape(loss=env.loss.loss)`` (plb/optimizer/solver.py:36): records the
    ``env.step`` / ``env.compute_loss`` calls made inside the ``with`` block and, on exit,
    replays their adjoints in reverse -- compute_loss_kernel_grad, the substep_grad's of the
    step, forward_kinematics.grad, set_velocity.grad -- with d(loss) = 1."""

    def __init__(self, env: TaichiEnv, after_step_grad=None):
        self.env = env
        self.events = []
        # optional hook ``after_step_grad(step, first_frame)``, called in the reverse sweep right after the adjoint
        # of an env step: the adjoint of that step's first frame is resident and d loss / d action[step] is final,
        # so a policy that produced action[step] from an observation of that frame can push its gradient back in
        # (Engine.add_frame_grad / add_primitive_grad) -- see optimizer/solver_nn.py
        self.after_step_grad = after_step_grad

    def __enter__(self):
        assert not self.env._is_copy, "gradients need tape mode: set_state(..., is_copy=False)"
        self.env._tape = self
        self.events = []
        self.env.loss.clear_loss()          # Tape zeroes the loss it differentiates
        return self

    def _record_step(self, first_frame):
        self.events.append(("step", first_frame, first_frame // self.env.simulator.substeps))

    def _record_loss(self, frame):
        self.events.append(("loss", frame))

    def __exit__(self, exc_type, exc, tb):
        self.env._tape = None
        if exc_type is not None:
            return False
        sim = self.env.simulator
        sim.grad_begin(sim.cur)
        for ev in reversed(self.events):
            if ev[0] == "loss":
                self.env.loss.compute_loss_kernel_grad(ev[1])
            else:
                sim.step_grad(ev[1], ev[2])
                if self.after_step_grad is not None:
                    self.after_step_grad(ev[2], ev[1])
        return False
