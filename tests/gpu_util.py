"""Helpers for the -m gpu tests: build an Engine for an oracle scene, pre-roll a state with the oracle."""
from __future__ import annotations

import numpy as np
import torch

from tests import emul
from tests.util import O


def engine_for(sim, prims, dtype="float64", max_frames=64, svd_grad_clamp=1e-6, **engine_kw):
    from plasticinelab_amd.engine.core import Engine
    plist = [dict(shape=p.shape, action_dim=p.action_dim, params=emul.prim_par(p), friction=p.friction,
                  action_scale=p.action_scale, lower_bound=p.lower_bound, upper_bound=p.upper_bound) for p in prims]
    return Engine(n_grid=sim.n_grid, n_particles=sim.n_particles, max_frames=max_frames, substeps=sim.substeps,
                  dt=sim.dt, p_vol=sim.p_vol, p_mass=sim.p_mass, gravity=sim.gravity,
                  ground_friction=sim.ground_friction, primitives=plist, dtype=dtype, svd_grad_clamp=svd_grad_clamp, **engine_kw)


def preroll(sim, prims, x0, actions, softness=666.0):
    """Run the oracle for len(actions) env steps from rest; returns (state, poses) as torch f64."""
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)
    with torch.no_grad():
        for a in actions:
            state, poses = O.env_step(sim, prims, softness, state, mats, poses, torch.as_tensor(a, dtype=O.DT))
    return state, mats, poses


def load_state(eng, frame, state, mats, poses, resort=True):
    x, v, C, F = [t.detach().numpy() for t in state]
    eng.set_frame(frame, x=x, v=v, F=F, C_=C, resort=resort)
    eng.set_materials(*[m.numpy() for m in mats])
    for k, pose in enumerate(poses):                # (pos, rot) or (pos, rot, gap)
        eng.set_primitive_state(k, frame, np.concatenate([t.detach().numpy().reshape(-1) for t in pose]))


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
