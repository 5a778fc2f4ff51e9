"""``torch.autograd.Function`` wrapper of a whole rollout: actions (H, A) -> total loss.

The backward pass is the hand-written adjoint of the HIP engine (no tracing): the forward
records the step / loss calls on a ``Tape`` and ``backward`` returns
``Primitives.get_grad(H)`` scaled by the incoming cotangent.  This is the counterpart of
``with ti.Tape(loss=env.loss.loss)`` + ``primitives.get_grad`` in the reference
(plb/optimizer/solver.py:31-44) and lets a torch policy / optimiser sit on top.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine.taichi_env import TaichiEnv, Tape


class RolloutLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, actions: torch.Tensor, env: TaichiEnv, sim_state, softness: float):
        acts = actions.detach().cpu().double().numpy()
        env.set_state(sim_state, softness, False)
        with Tape(env):
            for a in acts:
                env.step(a)
                env.compute_loss()
        grad = env.primitives.get_grad(len(acts))
        ctx.save_for_backward(torch.as_tensor(grad, dtype=actions.dtype, device=actions.device))
        return torch.as_tensor(env.loss.loss, dtype=actions.dtype, device=actions.device)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def rollout_loss(actions: torch.Tensor, env: TaichiEnv, sim_state=None, softness: float = 666.0):
    if sim_state is None:
        sim_state = env.get_state()["state"]
    return RolloutLoss.apply(actions, env, sim_state, softness)
