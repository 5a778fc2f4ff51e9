"""Mirror of plb.envs.env.PlasticineEnv and plb.envs.make
(/root/reference/plb/envs/env.py:12-86, plb/envs/__init__.py:16-20): Gym-style
reset()/step() over the MI355X engine.  ``gym`` is optional -- when it is not
installed the class still offers the same methods and minimal Box stand-ins.
"""
from __future__ import annotations

import os
import re

import numpy as np

from ..engine.taichi_env import TaichiEnv
from .scenes import ENV_NAMES, load_scene, load_variant_file

try:                                     # pragma: no cover - gym is absent in the build image
    import gym
    from gym.spaces import Box
    _Base = gym.Env
except Exception:                        # noqa: BLE001
    gym = None
    _Base = object

    class Box:                           # minimal stand-in
        def __init__(self, low, high, shape):
            self.low, self.high, self.shape = low, high, tuple(shape)

        def sample(self):
            return np.random.uniform(np.maximum(self.low, -1), np.minimum(self.high, 1), self.shape)


class PlasticineEnv(_Base):
    def __init__(self, cfg_path, version, nn=False, compute_dtype=None, target_grid=None):
        self.cfg_path = cfg_path
        cfg = self.load_varaints(cfg_path, version)
        if target_grid is not None:
            cfg.ENV.loss.target_path = ""
        else:
            tp = cfg.ENV.loss.target_path
            here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            if tp and not (os.path.exists(tp) or os.path.exists(os.path.join(here, tp))):
                # the reference ships its targets as plb/envs/assets/*.npy; they are not redistributed here
                raise FileNotFoundError(
                    f"target grid {tp!r} of {os.path.basename(cfg_path)} not found: pass make(..., assets_dir=<the reference's "
                    f"plb/envs/assets>) or make(..., target_grid=<(n,n,n) mass grid>)")
        self.taichi_env = TaichiEnv(cfg, nn, compute_dtype=compute_dtype)
        self.taichi_env.initialize()
        if target_grid is not None:
            self.taichi_env.loss.load_target_density(grids=target_grid)
        self.cfg = cfg.ENV
        self.taichi_env.set_copy(True)
        self._init_state = self.taichi_env.get_state()
        self._n_observed_particles = self.cfg.n_observed_particles
        self._max_episode_steps = 50                     # plb/envs/__init__.py:12
        obs = self.reset()
        self.observation_space = Box(-np.inf, np.inf, obs.shape)
        self.action_space = Box(-1, 1, (self.taichi_env.primitives.action_dim,))

    @property
    def unwrapped(self):
        return self

    def reset(self):                                     # env.py:28-31
        self.taichi_env.set_state(**self._init_state)
        self._recorded_actions = []
        return self._get_obs()

    def _get_obs(self, t=0):                             # env.py:33-41
        x = self.taichi_env.simulator.get_x(t)
        v = self.taichi_env.simulator.get_v(t)
        s = np.concatenate([p.get_state(t) for p in self.taichi_env.primitives])
        step_size = len(x) // self._n_observed_particles
        return np.concatenate((np.concatenate((x[::step_size], v[::step_size]), axis=-1).reshape(-1), s.reshape(-1)))

    def step(self, action):                              # env.py:43-57
        self.taichi_env.step(action)
        loss_info = self.taichi_env.compute_loss()
        self._recorded_actions.append(action)
        obs = self._get_obs()
        r = loss_info["reward"]
        if np.isnan(obs).any() or np.isnan(r):
            if np.isnan(r):
                print("nan in r")
            import datetime
            import pickle
            import tempfile
            import warnings
            # env.py:50-56: the episode's actions, for a replay.  The reference always has a real yml path to write next to; here
            # cfg_path may be a built-in scene name or sit in a read-only package directory -- whatever happens to the dump,
            # "NaN.." is the exception the caller sees.
            stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S_%f")
            base = os.path.basename(str(self.cfg_path)) or "scene"
            for where in (os.path.dirname(os.path.abspath(str(self.cfg_path))), tempfile.gettempdir()):
                try:
                    with open(os.path.join(where, f"{base}_nan_action_{stamp}"), "wb") as f:
                        pickle.dump(self._recorded_actions, f)
                    break
                except OSError as e:
                    warnings.warn(f"could not write the NaN action dump to {where}: {e}")
            raise Exception("NaN..")
        return obs, r, False, loss_info

    def render(self, mode="human"):
        return self.taichi_env.render(mode)

    @classmethod
    def load_varaints(cls, cfg_path, version):           # env.py:63-86 (name kept, typo included)
        assert version >= 1
        name = os.path.splitext(os.path.basename(cfg_path))[0]
        for builtin in ENV_NAMES:
            if builtin.lower() == name.lower() and not os.path.exists(cfg_path):
                return load_scene(builtin, version)
        return load_variant_file(cfg_path, version)


def make(env_name, nn=False, sdf_loss=10, density_loss=10, contact_loss=1, soft_contact_loss=False,
         compute_dtype=None, target_grid=None, assets_dir=None):
    """plb.envs.make: ``env_name`` like ``"Move-v1"``.  ``assets_dir`` is where the reference's
    ``envs/assets/*.npy`` targets live (they are not redistributed here); alternatively pass
    ``target_grid`` (an (n,n,n) mass grid)."""
    m = re.fullmatch(r"([A-Za-z]+)-v(\d+)", env_name)
    if not m:
        raise ValueError(f"bad env name {env_name!r}")
    name, version = m.group(1), int(m.group(2))
    if target_grid is None and assets_dir is not None:
        target_grid = np.load(os.path.join(assets_dir, f"{name}3D-v{version}.npy"))
    env = PlasticineEnv(f"{name.lower()}.yml", version, nn=nn, compute_dtype=compute_dtype, target_grid=target_grid)
    env.taichi_env.loss.set_weights(sdf=sdf_loss, density=density_loss, contact=contact_loss,
                                    is_soft_contact=soft_contact_loss)
    return env
