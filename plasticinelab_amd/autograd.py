"""``torch.autograd.Function`` wrappers of the engine: ONE FUNCTION PER ENV STEP.

    sess = RolloutSession(env)                     # set_state(..., is_copy=False): tape-mode frames
    carry = sess.begin()
    for k in range(H):
        obs = sess.observe(carry)                  # optional: what a policy sees (differentiable)
        action = policy(obs)                       # any torch code between the steps
        loss_k, carry = sess.step(action, carry)   # `substeps` substeps forward + Loss.compute_loss
    total = sum(losses)
    total.backward()                               # only now does the adjoint run, step by step, in reverse

The backward pass is the hand-written adjoint of the HIP engine (no tracing).  ``carry`` is a scalar token that makes
torch replay the steps in reverse order (the engine keeps the adjoint of only two particle frames, which is exactly
what a reverse sweep needs); ``EnvStep.backward`` runs ``Loss.compute_loss_kernel_grad`` of the step's last frame
(scaled by the cotangent of ``loss_k``), ``MPMSimulator.step_grad`` and returns d / d action.  This is the counterpart
of ``with ti.Tape(loss=env.loss.loss)`` + ``primitives.get_grad`` in the reference (plb/optimizer/solver.py:31-44),
with two differences a torch caller wants: nothing of the adjoint is paid for until ``backward`` is called, and the
caller's own differentiable ops may sit between the steps (a policy in the loop, per-step loss weights).

``rollout_loss(actions, env)`` is the one-call form: actions (H, A) -> total loss.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine.taichi_env import TaichiEnv


class RolloutSession:
    def __init__(self, env: TaichiEnv, sim_state=None, softness: float = 666.0, n_observed_particles: int = 200,
                 velocity_weight: float = 1.0):
        self.env, self.softness = env, float(softness)
        self.sim_state = env.get_state()["state"] if sim_state is None else sim_state
        self.n_obs, self.velocity_weight = int(n_observed_particles), float(velocity_weight)
        self.steps = 0                      # env steps taken forward
        self._pending = None                # index of the step whose backward must come next
        self._grad_started = False

    # ---- forward
    def begin(self, dtype=torch.float64, device="cpu") -> torch.Tensor:
        """Reset the engine to the episode start (tape mode) and return the first carry token."""
        self.env.set_state(self.sim_state, self.softness, False)
        self.env.loss.clear_loss()
        self.steps, self._pending, self._grad_started = 0, None, False
        return torch.zeros((), dtype=dtype, device=device, requires_grad=True)

    def step(self, action: torch.Tensor, carry: torch.Tensor):
        """-> (loss of this step as Loss.compute_loss reports it, next carry)."""
        return EnvStep.apply(action, carry, self)

    def observe(self, carry: torch.Tensor) -> torch.Tensor:
        """Observation of the current first frame, laid out like the reference's policy input
        (plb/engine/nn/mlp.py:63-84): x and velocity_weight * v of every k-th particle, then the manipulators' poses."""
        return Observe.apply(carry, self)

    # ---- helpers for the Functions
    def _obs_index(self):
        n = self.env.simulator.n_particles
        k = max(n // self.n_obs, 1)
        return np.arange(n // k) * k

    def _start_grad(self, k):
        if not self._grad_started:
            if k != self.steps - 1:
                raise RuntimeError(f"the reverse sweep starts at the last env step ({self.steps - 1}), got step {k}")
            self.env.simulator.grad_begin(self.env.simulator.cur)
            self._grad_started, self._pending = True, k
        if k != self._pending:
            raise RuntimeError(f"env steps are differentiated in reverse order: expected step {self._pending}, got {k} "
                               "(every step's carry must feed the next step)")


class EnvStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, action, carry, sess: RolloutSession):
        env, sim = sess.env, sess.env.simulator
        k = sess.steps
        if sess._grad_started:
            raise RuntimeError("this session has already been differentiated: call begin() for a new rollout")
        a = action.detach().cpu().double().numpy()
        first = sim.cur
        env.step(a)
        info = env.compute_loss()
        sess.steps += 1
        ctx.sess, ctx.k, ctx.first, ctx.last = sess, k, first, sim.cur
        ctx.meta = (action.dtype, action.device)
        ctx.set_materialize_grads(False)
        loss = torch.as_tensor(info["loss"], dtype=action.dtype, device=action.device)
        return loss, carry.detach() + 0.0

    @staticmethod
    def backward(ctx, g_loss, g_carry):
        sess, k = ctx.sess, ctx.k
        env, sim = sess.env, sess.env.simulator
        sess._start_grad(k)
        g = 0.0 if g_loss is None else float(g_loss)
        if g != 0.0:                                    # d total / d loss_k: the loss adjoint is linear in the weights
            w, soft = env.loss._weights, env.loss.soft_contact_loss
            env.loss.set_weights(w[0] * g, w[1] * g, w[2] * g, soft)
            try:
                env.loss.compute_loss_kernel_grad(ctx.last)
            finally:
                env.loss.set_weights(w[0], w[1], w[2], soft)
        sim.step_grad(ctx.first, k)
        sess._pending = k - 1
        grad = env.primitives.get_grad(k + 1)[k]
        dtype, device = ctx.meta
        return torch.as_tensor(grad, dtype=dtype, device=device), torch.zeros((), dtype=dtype, device=device), None


class Observe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, carry, sess: RolloutSession):
        sim = sess.env.simulator
        f = sim.cur
        idx = sess._obs_index()
        fr = sim.engine.get_frame(f, want=("x", "v"))
        part = np.concatenate([fr["x"][idx], fr["v"][idx] * sess.velocity_weight], axis=1).reshape(-1)
        prim = [p.get_state(f)[:7] for p in sess.env.primitives]
        ctx.sess, ctx.f, ctx.idx = sess, f, idx
        ctx.meta = (carry.dtype, carry.device)
        return torch.as_tensor(np.concatenate([part] + prim) if prim else part, dtype=carry.dtype, device=carry.device)

    @staticmethod
    def backward(ctx, g):
        sess, f, idx = ctx.sess, ctx.f, ctx.idx
        sim = sess.env.simulator
        # the adjoint of frame f is resident: the env step that starts there has just been differentiated -- or f is the
        # rollout's final frame (a terminal value / bootstrap on the last observation) and this is the first backward
        # call of the sweep, which then starts here (grad_begin zeroes the adjoints: it must come BEFORE anything is added)
        if not sess._grad_started:
            if f != sim.cur:
                raise RuntimeError(f"observation of frame {f} differentiated before the reverse sweep reached it (the sweep starts at frame {sim.cur})")
            sim.grad_begin(sim.cur)
            sess._grad_started, sess._pending = True, sess.steps - 1
        g = g.detach().cpu().double().numpy()
        n = sim.n_particles
        gp = g[:len(idx) * 6].reshape(len(idx), 6)
        xa, va = np.zeros((n, 3)), np.zeros((n, 3))
        xa[idx] = gp[:, :3]
        va[idx] = gp[:, 3:] * sess.velocity_weight
        sim.engine.add_frame_grad(f, xa=xa, va=va)
        base = len(idx) * 6
        for i in range(len(sess.env.primitives)):
            sim.engine.add_primitive_grad(i, f, g[base + 7 * i: base + 7 * i + 7])
        dtype, device = ctx.meta
        return torch.zeros((), dtype=dtype, device=device), None


def rollout_loss(actions: torch.Tensor, env: TaichiEnv, sim_state=None, softness: float = 666.0):
    """actions (H, A) -> total loss of the rollout (a torch scalar whose backward is the engine's adjoint)."""
    sess = RolloutSession(env, sim_state, softness)
    carry = sess.begin(dtype=actions.dtype, device=actions.device)
    total = None
    for k in range(actions.shape[0]):
        loss, carry = sess.step(actions[k], carry)
        total = loss if total is None else total + loss
    return total
