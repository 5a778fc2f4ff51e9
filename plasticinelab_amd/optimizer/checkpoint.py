"""Segment-checkpointed rollout gradient.

Same outputs as ``Solver.forward`` (total loss, d loss / d actions) but only ``segment`` env steps of particle
frames are resident at a time: the forward pass keeps one host checkpoint per segment, the backward pass re-runs a
segment's forward from its checkpoint, then walks it in reverse and hands the frame-0 adjoint over to the end of
the previous segment.  This is the schedule of the reference's ``plb/optimizer/long_term_gradient.ipynb`` (cells
2-4, ``forward2`` / ``copy_and_clear``), and what makes horizons x particle counts that exceed HBM (BASELINE
config 5: 1.5 GB per frame) differentiable.
"""
from __future__ import annotations

import numpy as np


def forward_checkpointed(env, init_state, actions, segment: int, softness: float = 666.0):
    """-> (loss, grad (H, action_dim)).  ``env`` is a TaichiEnv; needs ``max_steps > segment * substeps``."""
    sim, loss = env.simulator, env.loss
    eng = sim.engine
    H, T, sub = len(actions), int(segment), sim.substeps
    if T * sub >= sim.max_steps:
        raise ValueError(f"segment of {T} steps needs {T * sub + 1} frames, simulator has max_steps={sim.max_steps}")
    env.set_state(init_state, softness, False)          # sorts the storage order once ...
    # ... and it is kept for all segments: every segment reuses frames 0..T*sub, so a per-step re-sort would hand
    # out the same epoch numbers (and overwrite their permutations) while adjoints still carry those labels
    was_on = eng.set_resort(False)
    try:
        return _forward_checkpointed(env, init_state, actions, T, H, sub)
    finally:
        eng.set_resort(was_on)                          # as the owner had it (it may be off for an A/B run of its own)


def _forward_checkpointed(env, init_state, actions, T, H, sub):
    sim, loss = env.simulator, env.loss
    eng = sim.engine
    loss.clear_loss()
    checkpoints, total = {}, 0.0
    for i in range(H):
        if i % T == 0:
            if i > 0:
                total += loss.loss
                state = sim.get_state(T * sub)
                loss.clear_loss()
                sim.set_state(0, state, resort=False)
                sim.cur = 0
                checkpoints[i] = state
            else:
                checkpoints[0] = init_state
        env.step(actions[i])
        env.compute_loss()
    total += loss.loss

    sim.grad_begin(sim.cur)
    pieces, last = [], H
    for i in range(H - 1, -1, -1):
        f = (i % T) * sub
        loss.compute_loss_kernel_grad(f + sub)
        sim.step_grad(f, i % T)
        if i % T == 0:
            pieces.append(eng.get_action_grad(last - i))
            last = i
            if i > 0:
                start = i - T
                sim.set_state(0, checkpoints[start], resort=False)
                sim.cur = 0
                for s in range(start, i):                # re-run the earlier segment's forward
                    sim.step(False, actions[s])
                eng.segment_carry(0, T * sub)
    loss.loss = total
    return total, np.concatenate(pieces[::-1], axis=0)
