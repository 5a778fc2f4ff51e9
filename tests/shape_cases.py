"""The manipulator-shape rollout cases shared by tests/golden/make_golden.py (oracle side, writes
tests/golden/rollout_shapes.npz) and tests/test_gpu_shapes.py (HIP side): tilted movable Capsule / Torus driven
with 6-dof actions, the Rollingpin-v1 scene and the Chopsticks-v1 scene, all subsampled to 1500 particles."""
from __future__ import annotations

import numpy as np

N_PARTICLES = 1500
TARGET = "Move3D-v1"                 # any 64^3 target exercises the same loss code

_ACT6 = np.array([[0.9, 0.3, 0.1, 0.5, -0.4, 0.3, -0.9, 0.2, -0.1, -0.3, 0.6, 0.2],
                  [0.5, -0.2, 0.3, -0.6, 0.2, 0.4, -0.4, 0.1, 0.2, 0.3, -0.5, -0.2]])

CASES = {
    # name: (scene builder key, primitive override, soft contact loss, actions)
    "capsule_hard": ("tilted", ("Capsule", dict(h=0.06, r=0.03)), False, _ACT6),
    "capsule_soft": ("tilted", ("Capsule", dict(h=0.06, r=0.03)), True, _ACT6),
    "torus_hard": ("tilted", ("Torus", dict(tx=0.05, ty=0.02)), False, _ACT6),
    "torus_soft": ("tilted", ("Torus", dict(tx=0.05, ty=0.02)), True, _ACT6),
    # no reference task moves a Cylinder or a Box; SURVEY 8f rank 2 lists their adjoints all the same
    "cylinder_soft": ("tilted", ("Cylinder", dict(h=0.05, r=0.04)), True, _ACT6),
    "box_soft": ("tilted", ("Box", dict(size=(0.04, 0.03, 0.05))), True, _ACT6),
    "rollingpin": ("Rollingpin", None, True, np.array([[0.8, -0.5, -0.6], [0.6, 0.4, -0.3]])),
    # gap opened in step 1 so that the minimal-gap clamp is inactive, closed in steps 2-3
    "chopsticks": ("Chopsticks", None, True, np.array([[0.6, -0.2, -0.2, 0.5, -0.4, 0.3, -0.8],
                                                       [0.5, 0.1, 0.1, -0.3, 0.2, 0.4, 0.3],
                                                       [-0.4, 0.2, 0.2, 0.2, -0.1, -0.3, 0.3]])),
}


# every task family of the reference at its own v1 scene configuration (plb/envs/*.yml), subsampled; two env steps of
# seeded actions, soft contact loss
FAMILIES = ("Move", "TripleMove", "Rope", "Writer", "Torus", "Rollingpin", "Chopsticks", "Pinch", "Table", "Assembly")
_ADIM = {"Move": 6, "TripleMove": 18, "Rope": 6, "Writer": 3, "Torus": 3, "Rollingpin": 3, "Chopsticks": 7, "Pinch": 3, "Table": 3,
         "Assembly": 6}
for _i, _f in enumerate(FAMILIES):
    CASES[f"scene_{_f}"] = (_f, None, True, np.random.default_rng(100 + _i).uniform(-1, 1, (2, _ADIM[_f])) * 0.7)


def case_cfg(name):
    from plasticinelab_amd.envs.scenes import load_scene
    kind, override, soft, acts = CASES[name]
    if kind == "tilted":
        shape, kw = override
        cfg = load_scene("Move", 1)
        rot = np.array([0.9, 0.2, -0.3, 0.25]); rot /= np.linalg.norm(rot)
        prims = []
        for p in cfg.PRIMITIVES:
            d = {"shape": shape, "init_pos": tuple(p["init_pos"]), "init_rot": tuple(float(r) for r in rot), "friction": 0.9,
                 "action": {"dim": 6, "scale": (0.01,) * 6}}
            d.update(kw)
            prims.append(d)
        cfg["PRIMITIVES"] = prims
    else:
        cfg = load_scene(kind, 1)
    cfg.ENV.loss.target_path = ""
    return cfg, soft, acts


def subsample(x):
    k = len(x) // N_PARTICLES
    return np.ascontiguousarray(x[::k][:N_PARTICLES])
