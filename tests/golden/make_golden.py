"""Regenerates the fixtures under tests/golden/.  Run in the BUILD container only
(it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py shapes scenes targets rollout_small rollout_move rollout_shapes

What each fixture pins
  shapes.npz           the reference's own particle sampler (plb/engine/shapes/shape_maker.py, the one
                       reference file that runs here -- loaded by file path) for Move/TripleMove/Rope v1:
                       sha256 of the full float64 array + its first 64 rows.  REAL reference output.
  scenes.json          all 50 tasks (ten families x five versions, plb/envs/*.yml + VARIANTS): digest of the merged config tree as the
                       REAL YAML files give it, and sha256 / row count of the particle cloud the REAL reference sampler draws
                       for it.  Pins the built-in scene tables (plasticinelab_amd/envs/scenes.py) and "identical initial
                       conditions" for every task.  REAL reference data / output.
  target_<name>.npz    sparse copy (index + value of the non-zero nodes) of reference target mass grids
                       plb/envs/assets/<name>.npy, used as loss targets.  REAL reference data.
  target_sums.npz      sum / p_mass and non-zero count of all 50 reference target grids (KAT: every one
                       sums to exactly 10000 particle masses).  REAL reference data.
  rollout_*.npz        loss and d loss / d actions of Solver.forward-style rollouts computed by the
                       float64 ORACLE (oracle/plb_oracle.py), not by Taichi -- parity vs Taichi itself is
                       unpinned.
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def ref_shapes():
    spec = importlib.util.spec_from_file_location("ref_shape_maker", f"{REF}/plb/engine/shapes/shape_maker.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Shapes


def make_shapes():
    import yaml
    Shapes = ref_shapes()
    out = {}
    for name in ("move", "triplemove", "rope"):
        cfg = yaml.safe_load(open(f"{REF}/plb/envs/{name}.yml"))
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            x, _ = Shapes(cfg["SHAPES"]).get()
        x = np.ascontiguousarray(x, np.float64)
        out[f"{name}_sha256"] = np.array(hashlib.sha256(x.tobytes()).hexdigest())
        out[f"{name}_head"] = x[:64].copy()
        out[f"{name}_shape"] = np.array(x.shape)
        print(name, x.shape, out[f"{name}_sha256"])
    np.savez_compressed(os.path.join(HERE, "shapes.npz"), **out)


def make_scenes():
    """Every task of the reference (plb/envs/__init__.py:6-14): the scene as its YAML + variant give it (merged the way
    PlasticineEnv.load_varaints does, env.py:63-86) and the reference sampler's output for it."""
    import contextlib, io, json
    from plasticinelab_amd.envs.scenes import ENV_NAMES, load_variant_file
    from tests.util import canon_tree, scene_digest
    Shapes = ref_shapes()
    out = {}
    for name in ENV_NAMES:
        for version in range(1, 6):
            cfg = load_variant_file(f"{REF}/plb/envs/{name.lower()}.yml", version)
            shapes = [{k: (v if not isinstance(v, str) else v) for k, v in dict(s).items()} for s in cfg.SHAPES]
            with contextlib.redirect_stdout(io.StringIO()):
                x, _ = Shapes(shapes).get()
            x = np.ascontiguousarray(x, np.float64)
            out[f"{name}-v{version}"] = {"cfg_sha256": scene_digest(cfg), "x_sha256": hashlib.sha256(x.tobytes()).hexdigest(), "n": int(len(x))}
            print(f"{name}-v{version}", len(x), out[f"{name}-v{version}"]["x_sha256"][:12])
    with open(os.path.join(HERE, "scenes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def make_targets():
    names, sums, nnz = [], [], []
    p_mass = (1 / 64 * 0.5) ** 2
    for fn in sorted(os.listdir(f"{REF}/plb/envs/assets")):
        if not fn.endswith(".npy"):
            continue
        a = np.load(f"{REF}/plb/envs/assets/{fn}")
        names.append(fn[:-4]); sums.append(a.sum() / p_mass); nnz.append(int((a > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "target_sums.npz"), names=np.array(names), sums=np.array(sums), nnz=np.array(nnz))
    for name in ("Move3D-v1", "Rope3D-v1", "TripleMove3D-v1"):
        a = np.load(f"{REF}/plb/envs/assets/{name}.npy")
        idx = np.argwhere(a != 0).astype(np.int16)
        np.savez_compressed(os.path.join(HERE, f"target_{name}.npz"), n=np.array(a.shape[0]), idx=idx,
                            val=a[a != 0].astype(np.float64))
        print(name, idx.shape)


def _rollout(tag, scene, n_particles, H, action_fn, soft_contact=False):
    import ctypes
    import torch
    from tests.util import O, oracle_scene, sparse_target
    cfg, sim, prims, x0 = oracle_scene(scene, 1, n_particles=n_particles)
    tgt = sparse_target(f"{scene}3D-v1")
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libplb_oracle_c.so"))
    n = sim.n_grid
    sdf = np.empty((n, n, n)); npn = np.empty((n, n, n, 3))
    lib.plb_oracle_target_sdf.restype = ctypes.c_int
    lib.plb_oracle_target_sdf(np.ascontiguousarray(tgt).ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                              ctypes.c_double(sim.dx), ctypes.c_double(1000.0), ctypes.c_int(2 * n),
                              sdf.ctypes.data_as(ctypes.c_void_p), npn.ctypes.data_as(ctypes.c_void_p))
    A = sum(p.action_dim for p in prims)
    actions = action_fn(H, A)
    lcfg = O.LossCfg(soft_contact=soft_contact)
    t = time.time()
    L, g, states, poses, info = O.rollout_loss_and_grad(
        sim, lcfg, prims, 666.0, O.init_state(x0), O.materials(sim), O.init_poses(prims),
        torch.as_tensor(actions, dtype=O.DT), torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1)))
    print(tag, "loss", L, "time", time.time() - t)
    xf, vf, Cf, Ff = [s.numpy() for s in states[-1]]
    np.savez_compressed(os.path.join(HERE, f"rollout_{tag}.npz"), scene=np.array(scene), n_particles=np.array(len(x0)),
                        actions=actions, loss=np.array(L), grad=g.numpy(),
                        step_losses=np.array([[d["sdf_loss"], d["density_loss"], d["contact_loss"]] for d in info]),
                        x_final=xf, v_final=vf, F_final=Ff, C_final=Cf,
                        prim_final=np.array([np.concatenate([p.numpy(), r.numpy()]) for p, r in poses[-1]]),
                        soft_contact=np.array(soft_contact))


def make_rollout_small():
    def acts(H, A):
        a = np.random.default_rng(0).uniform(-1, 1, (H, A)) * 0.3
        a[:, 0] = 0.9; a[:, 3] = -0.9      # drive both manipulators into the ball
        return a
    _rollout("small", "Move", 2000, 3, acts)
    _rollout("small_soft", "Move", 2000, 2, acts, soft_contact=True)


def make_rollout_move():
    from tests.util import seeded_actions
    _rollout("move_v1", "Move", None, 50, lambda H, A: seeded_actions(H, A, seed=0, scale=0.01))


def make_rollout_shapes():
    """Oracle loss / action gradient / final manipulator state for tests/shape_cases.py."""
    import ctypes
    import torch
    from tests.util import O, oracle_prims, sparse_target
    from tests.shape_cases import CASES, TARGET, case_cfg, subsample
    from plasticinelab_amd.engine.shapes import Shapes
    tgt = sparse_target(TARGET)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libplb_oracle_c.so"))
    n = tgt.shape[0]
    sdf = np.empty((n, n, n)); npn = np.empty((n, n, n, 3))
    lib.plb_oracle_target_sdf.restype = ctypes.c_int
    lib.plb_oracle_target_sdf(np.ascontiguousarray(tgt).ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                              ctypes.c_double(1.0 / n), ctypes.c_double(1000.0), ctypes.c_int(2 * n),
                              sdf.ctypes.data_as(ctypes.c_void_p), npn.ctypes.data_as(ctypes.c_void_p))
    out = {}
    for name in CASES:
        cfg, soft, acts = case_cfg(name)
        x0 = subsample(Shapes(cfg.SHAPES).get()[0])
        prims = oracle_prims(cfg)
        s = cfg.SIMULATOR
        sim = O.SimCfg(n_particles=len(x0), yield_stress=s.yield_stress, E=s.E, nu=s.nu, ground_friction=s.ground_friction,
                       gravity=tuple(s.gravity))
        t = time.time()
        L, g, states, poses, _ = O.rollout_loss_and_grad(
            sim, O.LossCfg(soft_contact=soft), prims, 666.0, O.init_state(x0), O.materials(sim), O.init_poses(prims),
            torch.as_tensor(acts, dtype=O.DT), torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1)))
        print(name, "loss", L, "time", time.time() - t)
        out[f"{name}_actions"] = acts
        out[f"{name}_loss"] = np.array(L)
        out[f"{name}_grad"] = g.numpy()
        out[f"{name}_x_final"] = states[-1][0].numpy()
        out[f"{name}_prim_final"] = np.array([np.concatenate([t.numpy().reshape(-1) for t in po]) for po in poses[-1]])
    np.savez_compressed(os.path.join(HERE, "rollout_shapes.npz"), **out)


def _target_sdf(tgt, dx):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libplb_oracle_c.so"))
    n = tgt.shape[0]
    sdf = np.empty((n, n, n)); npn = np.empty((n, n, n, 3))
    lib.plb_oracle_target_sdf.restype = ctypes.c_int
    lib.plb_oracle_target_sdf(np.ascontiguousarray(tgt).ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                              ctypes.c_double(dx), ctypes.c_double(1000.0), ctypes.c_int(2 * n),
                              sdf.ctypes.data_as(ctypes.c_void_p), npn.ctypes.data_as(ctypes.c_void_p))
    return sdf


def make_gym():
    """The Gym surface (plb/envs/env.py:28-57) on Move-v1 as the ORACLE computes it: 50 copy-mode env steps with
    seeded actions; per step the reward (loss.py:288-298: start_loss - step loss), the loss terms and the IoU, and
    the 1214-long observation (200 particles x (x, v) + 2 x 7 manipulator state) after steps 1, 25 and 50.
    Softness 0: the reference's Gym path never calls Primitives.set_softness (only the solvers do, solver.py:33), so
    its manipulators collide with the field's initial value 0 -- hard contact (primive_base.py:93-94: dist <= 0)."""
    import torch
    from tests.util import O, oracle_scene, sparse_target
    cfg, sim, prims, x0 = oracle_scene("Move", 1)
    tgt = sparse_target("Move3D-v1")
    sdf = _target_sdf(tgt, sim.dx)
    td, ts = torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(sdf.reshape(-1))
    H, A = 50, sum(p.action_dim for p in prims)
    actions = np.random.default_rng(7).uniform(-1, 1, (H, A)) * 0.6
    lcfg = O.LossCfg()
    state, mats, poses = O.init_state(x0), O.materials(sim), O.init_poses(prims)

    def obs(state, poses):
        x, v = state[0].numpy(), state[1].numpy()
        k = len(x) // 200
        s = np.concatenate([np.concatenate([t.numpy().reshape(-1) for t in po]) for po in poses])
        return np.concatenate((np.concatenate((x[::k], v[::k]), axis=-1).reshape(-1), s))

    out = {"actions": actions, "obs_0": obs(state, poses)}
    with torch.no_grad():
        l0, parts0 = O.compute_loss(sim, lcfg, prims, state[0], poses, td, ts)
        out["start_loss"] = np.array(float(l0))
        out["init_iou"] = np.array(float(O.iou(parts0["grid_m"], td)))
        rewards, terms = [], []
        t = time.time()
        for i in range(H):
            state, poses = O.env_step(sim, prims, 0.0, state, mats, poses, torch.as_tensor(actions[i], dtype=O.DT))
            l, parts = O.compute_loss(sim, lcfg, prims, state[0], poses, td, ts)
            rewards.append(float(l0) - float(l))
            terms.append([float(l), float(parts["sdf_loss"]), float(parts["density_loss"]), float(parts["contact_loss"]),
                          float(O.iou(parts["grid_m"], td))])
            if i + 1 in (1, 25, 50):
                out[f"obs_{i + 1}"] = obs(state, poses)
        print("gym rollout", time.time() - t, "s; reward[0], reward[-1] =", rewards[0], rewards[-1])
    out["rewards"] = np.array(rewards)
    out["terms"] = np.array(terms)
    np.savez_compressed(os.path.join(HERE, "gym_move_v1.npz"), **out)


if __name__ == "__main__":
    for what in sys.argv[1:]:
        globals()[f"make_{what}"]()
