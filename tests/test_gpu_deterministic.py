"""-m gpu: plmpm_config.deterministic = 1 -- integer-limb accumulation instead of floating-point atomics
(plasticinelab_amd/csrc/plmpm_kernels.h: det_add).  The same rollout must give the same BITS every time: across repeated
runs of one engine, across two engines, across re-sorts, for both scalar types and both contact losses -- and it must
still be the rollout of the oracle (same tolerances as tests/test_gpu_rollout.py)."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN, sparse_target
from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu


def make_env(dtype, soft_contact, deterministic=True):
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    cfg = load_scene("Move", 1)
    cfg.ENV.loss.target_path = ""
    cfg.SIMULATOR["deterministic"] = deterministic
    env = TaichiEnv(cfg, compute_dtype=dtype)
    env.initialize()
    env.loss.load_target_density(grids=sparse_target("Move3D-v1"))
    env.loss.set_weights(10, 10, 1, soft_contact)
    return env


def rollout(env, actions, state0):
    from plasticinelab_amd.optimizer.solver import Solver
    loss, grad = Solver(env, None, None, softness=666.0, horizon=len(actions)).forward(state0, actions)
    sim = env.simulator
    fr = sim.engine.get_frame(sim.cur)
    return np.float64(loss), np.array(grad), fr["x"].copy(), fr["v"].copy(), fr["F"].copy()


def same_bits(a, b):
    return all(np.ascontiguousarray(x).tobytes() == np.ascontiguousarray(y).tobytes() for x, y in zip(a, b))


@pytest.mark.parametrize("soft", [False, True])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_move_v1_rollout_is_bit_reproducible(dtype, soft):
    """Move-v1 (64^3, 10k particles, 2 manipulators), 12 env steps x 19 substeps forward + reverse, re-sorts inside."""
    g = np.load(os.path.join(GOLDEN, "rollout_move_v1.npz"))
    actions = g["actions"][:12]
    env = make_env(dtype, soft)
    assert env.simulator.engine.deterministic
    state0 = env.get_state()["state"]
    runs = [rollout(env, actions, state0) for _ in range(3)]
    env2 = make_env(dtype, soft)
    runs.append(rollout(env2, actions, env2.get_state()["state"]))
    for r in runs[1:]:
        assert same_bits(runs[0], r), "deterministic engine: two runs of the same rollout differ"
    # ... and it is the rollout the fp-atomics engine computes, to that engine's own round-off
    ref = rollout(make_env(dtype, soft, deterministic=False), actions, state0)
    ltol, gtol = (1e-11, 1e-8) if dtype == "float64" else (1e-5, 1e-4)
    print(f"\n[{dtype} soft={soft}] loss {runs[0][0]:.15g}; vs fp atomics: loss rel {abs(runs[0][0] - ref[0]) / abs(ref[0]):.2e}, "
          f"grad max-rel {relerr(runs[0][1], ref[1]):.2e}")
    assert abs(runs[0][0] - ref[0]) / abs(ref[0]) < ltol
    assert relerr(runs[0][1], ref[1]) < gtol


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_deterministic_engine_matches_oracle(dtype):
    g = np.load(os.path.join(GOLDEN, "rollout_move_v1.npz"))
    env = make_env(dtype, False)
    loss, grad, *_ = rollout(env, g["actions"], env.get_state()["state"])
    lerr, gerr = abs(loss - float(g["loss"])) / abs(float(g["loss"])), relerr(grad, g["grad"])
    print(f"\n[move_v1 deterministic {dtype}] loss rel {lerr:.3e}; grad max-rel err {gerr:.3e}")
    ltol, gtol = (1e-9, 1e-6) if dtype == "float64" else (1e-5, 1e-4)
    assert lerr < ltol and gerr < gtol


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_small_deterministic_rollout_repeats_and_matches_oracle(dtype):
    """The 3-step / 2000-particle golden rollout on the integer-limb engine: the oracle's numbers (tolerances of test_gpu_rollout), and
    the same bits when the rollout is run again.  (Small enough for the CPU interpreter of the device source, tests/test_emul_tier.py.)"""
    from tests.test_gpu_rollout import make_env_sub
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    env = make_env_sub("Move", int(g["n_particles"]), dtype, deterministic=True)
    assert env.simulator.engine.deterministic
    state0 = env.get_state()["state"]
    a = rollout(env, g["actions"], state0)
    b = rollout(env, g["actions"], state0)
    assert same_bits(a, b)
    ltol, gtol = (1e-10, 1e-7) if dtype == "float64" else (1e-5, 1e-4)
    assert abs(a[0] - float(g["loss"])) / abs(float(g["loss"])) < ltol and relerr(a[1], g["grad"]) < gtol


def test_large_cloud_is_bit_reproducible():
    """128^3, 200k particles, two manipulators pressing on the cube (the benchmark workload, smaller): every LDS-tile
    path of the normal engine is replaced by limb atomics here; two engines, identical bits."""
    import torch
    import bench
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    cfg = bench.workload_cfg(200_000, 2, max_steps=2 * 39 + 1)
    cfg.SIMULATOR["deterministic"] = True
    acts = bench.seeded_actions(2, 6)
    out = []
    for _ in range(2):
        env = TaichiEnv(cfg, compute_dtype="float32", device=torch.device("cuda", 0))
        env.initialize()
        env.loss.load_target_density(grids=bench._target(env.init_particles, env.simulator))
        env.loss.set_weights(10, 10, 1, False)
        state0 = env.get_state()["state"]
        out.append(rollout(env, acts, state0))
        out.append(rollout(env, acts, state0))
        env.simulator.engine.close()
    for r in out[1:]:
        assert same_bits(out[0], r)
    assert np.isfinite(out[0][0]) and np.abs(out[0][1]).max() > 0


def test_deterministic_slab_ranks_are_bit_reproducible(tmp_path):
    """Two and three z-slab ranks (gloo on the one GPU) with migration every env step, the rollout run twice: every
    rank's loss, action gradient and final particles are the same bits both times -- the leaving rows are packed in
    slot order, so the arrival order (which tie-breaks the neighbour's stable re-sort) is fixed too."""
    from tests.test_gpu_distributed import launch
    H = 6
    acts = np.zeros((H, 6))
    acts[:, 2] = 0.9; acts[:, 5] = 0.9              # both spheres push +z: rows cross the faces
    acts[:, 0] = 0.5; acts[:, 3] = -0.5
    acts += np.random.default_rng(4).uniform(-0.1, 0.1, acts.shape)
    for world in (2, 3):
        runs = []
        for k in range(2):
            d = tmp_path / f"w{world}_{k}"
            d.mkdir()
            runs.append(launch(d, world, "float32", acts, 10, 1, deterministic=True))
        assert sum(int(r["rows_moved"]) for r in runs[0]) > 0
        for a, b in zip(*runs):
            for key in ("loss", "grad", "ids", "x", "v"):
                assert np.ascontiguousarray(a[key]).tobytes() == np.ascontiguousarray(b[key]).tobytes(), (world, key)
