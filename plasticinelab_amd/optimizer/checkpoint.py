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
    if hasattr(eng, "reenter"):                         # a z-slab rank: its population changes with every migration
        return _forward_checkpointed_slab(env, init_state, actions, T, H, sub, softness)
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


def _forward_checkpointed_slab(env, init_state, actions, T, H, sub, softness):
    """The same schedule on a z-slab rank (plasticinelab_amd.distributed.SlabEngine; every rank calls this with the same
    arguments -- the migrations and halo exchanges inside are collective).

    A rank's rows change with every migration, so a checkpoint is the POPULATION at a segment boundary -- global ids,
    state and materials of the rows the rank holds there (``SlabEngine.checkpoint``) -- and a segment re-enters the engine
    with it as a new episode (``reenter``: ``plmpm_set_population`` + ids + frame 0 + materials).  The reverse sweep of a
    segment ends at its frame 0 in the rows left by the migration its first step began with; ``adjoint_to_reentry_rows``
    sends those adjoint rows home, which puts the adjoint into the checkpoint's rows.  It is handed to the end of the
    earlier segment BY GLOBAL ID (the re-run of that segment may store its final frame in another order), the
    manipulators' pose adjoints with it.  BASELINE config 5 -- 512^3 / 16M particles, 60 GiB of per-frame grids per rank
    for ONE env step -- becomes differentiable over a rollout this way."""
    sim, loss = env.simulator, env.loss
    eng = sim.engine
    env.set_state(init_state, softness, False)
    first_ck = dict(eng.checkpoint(0), prims=[p.get_state(0) for p in env.primitives])

    def restore(ck, collective=True):
        eng.reenter(ck, collective=collective)
        sim.n_particles = env.n_particles = len(ck["ids"])          # Observe / get_state size their arrays by these
        for st, p in zip(ck["prims"], env.primitives):
            p.set_state(0, st)
        sim.cur = 0

    # Whatever happens, leave the engine with the population -- and the row counts the callers read -- it started with.  The
    # full restore re-synchronises the device-side exchange, which is a collective: it may only run where EVERY rank runs it,
    # i.e. after a normal return or after a failure the ranks agreed on (SlabEngine._agree / _check mark those exceptions
    # `collective`).  A failure of this rank alone (NaN guard, engine error, out of memory) restores local state only: a
    # collective here would pair with whatever collective the other ranks are in and hang them or corrupt their reduction.
    try:
        out = _checkpointed_slab_sweeps(env, actions, T, H, sub, first_ck, restore)
    except BaseException as exc:
        restore(first_ck, collective=bool(getattr(exc, "collective", False)))
        raise
    restore(first_ck)
    return out


def _checkpointed_slab_sweeps(env, actions, T, H, sub, first_ck, restore):
    sim, loss = env.simulator, env.loss
    eng = sim.engine
    loss.clear_loss()
    checkpoints, total = {0: first_ck}, 0.0
    for i in range(H):
        if i % T == 0 and i > 0:
            total += loss.loss
            ck = dict(eng.checkpoint(T * sub), prims=[p.get_state(T * sub) for p in env.primitives])
            loss.clear_loss()
            restore(ck)
            checkpoints[i] = ck
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
                eng.adjoint_to_reentry_rows(0)
                ids = checkpoints[i]["ids"]
                ga = eng.get_frame_grad(0)                          # rows of checkpoints[i], in its order
                pg = [eng.get_primitive_grad(k, 0) for k in range(len(env.primitives))]
                start = i - T
                restore(checkpoints[start])
                for s in range(start, i):                           # re-run the earlier segment's forward
                    sim.step(False, actions[s])
                sim.grad_begin(T * sub)
                now = eng.get_ids(T * sub)
                # a face particle that changed sides concerns the two ranks of that face only; the others would walk on into
                # their next collective and hang.  So the verdict is agreed on first and every rank raises together.
                err = None
                if len(now) != len(ids) or not np.array_equal(np.sort(now), np.sort(ids)):
                    err = RuntimeError("the re-run of a segment ended with another set of rows on this rank than the run it was checkpointed "
                                       "from (a particle on a slab face changed sides by round-off): use cfg.SIMULATOR.deterministic")
                eng._agree(err, "the re-run of a checkpointed segment")
                order = np.argsort(ids, kind="stable")[np.searchsorted(np.sort(ids), now)]       # row of `ids` holding each id of `now`
                eng.add_frame_grad(T * sub, xa=ga["x"][order], va=ga["v"][order], Fa=ga["F"][order], Ca=ga["C"][order])
                for k, g in enumerate(pg):
                    eng.add_primitive_grad(k, T * sub, g)
    loss.loss = total
    return total, np.concatenate(pieces[::-1], axis=0)
