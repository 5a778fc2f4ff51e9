"""-m gpu: the Gym surface the north star keeps unchanged -- plb.envs.make / PlasticineEnv.reset / step / _get_obs
(/root/reference/plb/envs/env.py:28-57, plb/envs/__init__.py:16-20) -- on Move-v1, against the oracle's copy-mode
rollout (tests/golden/gym_move_v1.npz, made by tests/golden/make_golden.py gym): the 1214-long observation
(200 particles x (x, v) + 2 manipulators x 7), 50 step()s with their rewards, reset() restoring the episode start,
the NaN guard, and the per-primitive queries of SURVEY 8(b) (Primitive.sdf / set_velocity / min_dist / dist_norm)."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN, O, oracle_scene, sparse_target
from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu


def make_env(dtype):
    from plasticinelab_amd.envs import make
    return make("Move-v1", compute_dtype=dtype, target_grid=sparse_target("Move3D-v1"))


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 1e-4)])
def test_gym_episode_matches_oracle(dtype, tol):
    g = np.load(os.path.join(GOLDEN, "gym_move_v1.npz"))
    env = make_env(dtype)
    assert env.action_space.shape == (6,) and env.observation_space.shape == (1214,)
    obs = env.reset()
    assert obs.shape == (1214,) and obs.dtype == np.float64
    assert np.abs(obs - g["obs_0"]).max() < 1e-12                   # frame 0 is the caller's float64 state either way
    rewards = []
    for i, a in enumerate(g["actions"]):
        obs, r, done, info = env.step(a)
        assert done is False and set(info) >= {"reward", "loss", "iou", "incremental_iou", "sdf_loss", "density_loss", "contact_loss"}
        assert r == info["reward"]
        rewards.append(r)
        if i + 1 in (1, 25, 50):
            ref = g[f"obs_{i + 1}"]
            # positions and manipulator poses; velocities (entries 3:6 of every particle row) are small numbers
            assert np.abs(obs - ref).max() < tol * max(1.0, np.abs(ref).max()), (i + 1, np.abs(obs - ref).max())
            assert abs(info["loss"] - g["terms"][i, 0]) <= tol * abs(g["terms"][i, 0])
            assert abs(info["iou"] - g["terms"][i, 4]) <= max(tol, 1e-9) * max(abs(g["terms"][i, 4]), 1e-3)
    rewards = np.array(rewards)
    # reward = start_loss - loss of the step (loss.py:288-298): same sign and size as the oracle's, step by step
    scale = np.abs(g["start_loss"])
    assert np.abs(rewards - g["rewards"]).max() < tol * scale
    assert (np.sign(rewards[np.abs(g["rewards"]) > 10 * tol * scale]) == np.sign(g["rewards"][np.abs(g["rewards"]) > 10 * tol * scale])).all()
    assert len(env._recorded_actions) == 50
    # reset() restores the episode start exactly, and the episode replays
    obs0 = env.reset()
    assert np.abs(obs0 - g["obs_0"]).max() < 1e-12 and env._recorded_actions == []
    _, r1, _, _ = env.step(g["actions"][0])
    assert abs(r1 - rewards[0]) <= 1e-12 * scale + (0 if dtype == "float64" else 1e-6 * scale)


def test_nan_guard_raises(tmp_path):
    """env.py:43-57: NaN in the observation or the reward raises, after the episode's actions were pickled next to the
    scene file (`<cfg_path>_nan_action_<timestamp>`) for a replay."""
    import glob
    import pickle
    env = make_env("float32")
    env.cfg_path = str(tmp_path / "move.yml")                        # where the dump goes
    env.reset()
    env.step(np.full(6, 0.25))
    st = env.taichi_env.get_state()
    st["state"][1][7, 1] = np.nan                                    # one particle's velocity
    env.taichi_env.set_state(**st)
    with pytest.raises(Exception, match="NaN"):
        env.step(np.zeros(6))
    dumps = glob.glob(str(tmp_path / "move.yml_nan_action_*"))
    assert len(dumps) == 1
    acts = pickle.load(open(dumps[0], "rb"))
    assert len(acts) == 2 and np.allclose(acts[0], 0.25) and np.allclose(acts[1], 0.0)


def test_simulator_state_fields():
    """mpm_simulator.py:35-38: x / v / C / F read as fields (`sim.x[f, i]`, `sim.x[f]`, `.to_numpy()`), as loss.py:18-19 and
    the notebook do."""
    env = make_env("float64")
    env.reset()
    sim = env.taichi_env.simulator
    x0 = sim.get_x(0)
    assert np.array_equal(sim.x[0], x0) and np.array_equal(sim.x[0, 5], x0[5]) and sim.x[0, 5, 1] == x0[5, 1]
    assert sim.v[0].shape == (len(x0), 3) and sim.F[0].shape == (len(x0), 3, 3) and np.allclose(sim.F[0, 3], np.eye(3))
    assert sim.x.to_numpy().shape == (sim.cur + 1, len(x0), 3)


def test_primitive_queries():
    """Primitive.sdf (primive_base.py:57-60), set_velocity (:184-192), min_dist (loss.py:123-128) through the C ABI."""
    import torch
    env = make_env("float64")
    env.reset()
    te = env.taichi_env
    cfg, sim, prims, x0 = oracle_scene("Move", 1)
    pts = np.random.default_rng(0).uniform(0.3, 0.9, (64, 3))
    for k, p in enumerate(te.primitives):
        st = p.get_state(0)
        ref = O.prim_sdf(prims[k], torch.as_tensor(st[:3]), torch.as_tensor(st[3:7]), torch.as_tensor(pts)).numpy()
        got = p.sdf(0, pts)
        assert got.shape == (64,) and np.abs(got - ref).max() < 1e-12
        assert abs(p.sdf(0, pts[3]) - ref[3]) < 1e-12
    # hard contact loss: min_dist[None] is the smallest clamped distance of any particle to the primitive
    te.compute_loss()
    x = te.simulator.get_x(0)
    for k, p in enumerate(te.primitives):
        d = np.maximum(p.sdf(0, x), 0.0).min()
        assert abs(p.min_dist[None] - d) < 1e-12 and float(p.dist_norm) == 0.0
    # set_velocity re-derives the step's manipulator velocities from the stored action: stepping after it is unchanged
    a = np.array([0.5, -0.2, 0.1, -0.4, 0.3, 0.2])
    te.step(a)
    x1 = te.simulator.get_x(0).copy()
    env.reset()
    te.primitives.set_action(0, te.simulator.substeps, a)
    for p in te.primitives:
        p.set_velocity(0, te.simulator.substeps)
    te.simulator.step(is_copy=True, action=None)
    assert np.abs(te.simulator.get_x(0) - x1).max() < 1e-13


def test_make_names_the_missing_assets():
    from plasticinelab_amd.envs import make
    with pytest.raises(FileNotFoundError, match="assets_dir"):
        make("Move-v1")
    with pytest.raises(ValueError):
        make("Move_v1")
