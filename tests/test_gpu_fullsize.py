"""-m gpu: the BASELINE config-3 workload at FULL size (128^3 grid, 500k particles, two manipulators) -- too big for
the oracle, so parity goes through size-independent properties:

* mass: the loss's mass scatter sums to exactly N * p_mass;
* momentum: one substep in free fall with no manipulator in reach changes the total momentum by m g dt;
* the hand-derived adjoint against central finite differences of the loss along a random action direction
  (float64 engine, two env steps = 78 substeps fwd + bwd, across the storage re-sort cadence);
* fp32 engine against the fp64 engine on the same rollout: loss and action gradient within the north-star 1e-4.
"""
import numpy as np
import pytest

from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu


def build(dtype, monkeypatch=None, resort=None):
    import torch
    import bench
    if monkeypatch is not None and resort is not None:
        monkeypatch.setenv("PLMPM_RESORT_STEPS", str(resort))

    class A:
        particles, quality, steps, warmup = 500_000, 2, 2, 0
    A.dtype = dtype
    env, _ = bench.build_env(A, torch.device("cuda", 0))
    return env, bench


def test_mass_and_momentum_full_size():
    env, bench = build("float32")
    sim = env.simulator
    eng = sim.engine
    gm = eng.grid_mass(0)
    assert abs(gm.sum() / (sim.n_particles * sim.p_mass) - 1.0) < 1e-5
    nodes, blocks = eng.grid_stats(0)
    assert nodes == int((gm > 0).sum()) and blocks > 0
    # one substep from rest: every particle (all clear of the floor and of the manipulators' reach is not required:
    # the spheres start outside the cube) gains dt * g * 30 in v (grid_op's gravity term)
    st = env.get_state()["state"]
    env.set_state(st, 666.0, False)
    eng.substep(0)
    v1 = eng.get_frame(1, want=("v",))["v"]
    g = np.asarray(sim.default_gravity, float) * 30.0 * sim.dt
    far = np.ones(len(v1), bool)
    for p in env.primitives:
        c = np.asarray(p.cfg.init_pos, float)
        far &= np.linalg.norm(st[0] - c, axis=1) > 0.12
    assert far.sum() > 100_000
    assert np.abs(v1[far] - g).max() < 1e-6 * max(np.abs(g).max(), 1e-9) + 2e-7


def test_adjoint_matches_finite_differences_full_size(monkeypatch):
    env, bench = build("float64", monkeypatch, resort=1)          # re-sort before the second env step
    A = env.primitives.action_dim
    acts = bench.seeded_actions(2, A)
    st = env.get_state()["state"]

    def run(a, grad):
        env.set_state(st, 666.0, False)
        if grad:
            loss = bench.rollout(env, a)
            return loss, env.primitives.get_grad(len(a))
        env.loss.clear_loss()
        for ai in a:
            env.step(ai)
            env.compute_loss()
        return env.loss.loss, None

    loss, g = run(acts, True)
    d = np.random.default_rng(1).standard_normal(acts.shape)
    d /= np.abs(d).max()
    eps = 1e-5                                                      # FD error 2e-4 at 1e-4, 3e-5 from 3e-5 down
    lp, _ = run(acts + eps * d, False)
    lm, _ = run(acts - eps * d, False)
    fd = (lp - lm) / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd) > 0 and abs(an - fd) / abs(fd) < 1e-4, (an, fd)


def test_fp32_engine_tracks_fp64_engine_full_size():
    out = {}
    for dtype in ("float64", "float32"):
        env, bench = build(dtype)
        acts = bench.seeded_actions(2, env.primitives.action_dim)
        st = env.get_state()["state"]
        env.set_state(st, 666.0, False)
        loss = bench.rollout(env, acts)
        out[dtype] = (loss, env.primitives.get_grad(2).copy())
        del env
    l64, g64 = out["float64"]
    l32, g32 = out["float32"]
    assert abs(l32 - l64) / abs(l64) < 1e-4
    assert relerr(g32, g64) < 1e-4
