"""CPU tier: host-side pieces that define "identical initial conditions" -- the particle sampler
(pinned to REAL reference output, tests/golden/shapes.npz), scene registry, config tree."""
import hashlib
import os

import numpy as np
import pytest

from tests.util import GOLDEN
from plasticinelab_amd.config import CfgNode, as_value, get_cfg_defaults, merge_lists
from plasticinelab_amd.engine.shapes import Shapes
from plasticinelab_amd.envs.scenes import load_scene


@pytest.mark.parametrize("scene,key", [("Move", "move"), ("TripleMove", "triplemove"), ("Rope", "rope")])
def test_sampler_is_bit_identical_to_reference(scene, key):
    g = np.load(os.path.join(GOLDEN, "shapes.npz"))
    x, colors = Shapes(load_scene(scene, 1).SHAPES).get()
    x = np.ascontiguousarray(x, np.float64)
    assert tuple(g[f"{key}_shape"]) == x.shape
    assert np.array_equal(x[:64], g[f"{key}_head"])                       # bit exact
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g[f"{key}_sha256"])
    assert len(colors) == len(x)


def test_sampler_restores_global_rng_state():
    np.random.seed(123)
    a = np.random.random()
    np.random.seed(123)
    Shapes(load_scene("Move", 1).SHAPES)
    assert np.random.random() == a


def test_scene_constants():
    cfg = load_scene("Move", 1)
    assert cfg.SIMULATOR.n_particles == 10000 and cfg.SIMULATOR.yield_stress == 200.0
    assert len(cfg.PRIMITIVES) == 2 and cfg.PRIMITIVES[0]["action"]["dim"] == 3
    rope = load_scene("Rope", 3)
    assert rope.SIMULATOR.ground_friction == 0.3 and rope.PRIMITIVES[2]["shape"] == "Cylinder"
    assert rope.PRIMITIVES[2]["init_pos"][0] == 0.48953026610561057
    with pytest.raises(KeyError):
        load_scene("Nope", 1)
    with pytest.raises(ValueError):
        load_scene("Move", 6)


def test_cfg_tree_semantics():
    cfg = get_cfg_defaults()
    assert cfg.SIMULATOR.dtype == "float64" and cfg.ENV.loss.weight.sdf == 10
    with pytest.raises(KeyError):
        cfg.merge({"SIMULATOR": {"not_a_key": 1}}, strict=True)
    cfg.merge({"SIMULATOR": {"E": 1.0}})
    assert cfg.SIMULATOR.E == 1.0 and cfg.SIMULATOR.nu == 0.2
    assert as_value("(0.5, 1/4)") == (0.5, 0.25) and as_value("127<<16") == 127 << 16 and as_value("abc") == "abc"
    merged = merge_lists([{"a": 1, "b": {"c": 2}}, {"a": 5, "b": {"c": 6}}], [{"b": {"c": 3}}])
    assert merged == [{"a": 1, "b": {"c": 3}}, {"a": 5, "b": {"c": 6}}]
    with pytest.raises(ValueError):
        merge_lists([{"a": 1}], [{"zz": 1}])
    assert isinstance(CfgNode({"x": {"y": 1}}).x, CfgNode)


def _scene_fixture():
    import json
    with open(os.path.join(GOLDEN, "scenes.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["Move", "TripleMove", "Rope", "Writer", "Torus", "Rollingpin", "Chopsticks", "Pinch", "Table", "Assembly"])
def test_every_task_matches_the_reference_scene_files(name):
    """All 50 tasks (ten families x versions 1-5): the built-in scene tables give the config tree the reference's YAML + VARIANTS
    give (digest of the canonical tree, taken from the REAL files: tests/golden/scenes.json), and the sampler draws the particle
    cloud the REAL reference sampler draws for it, bit for bit -- "identical initial conditions" for every task, not only Move."""
    from tests.util import scene_digest
    fx = _scene_fixture()
    for version in range(1, 6):
        cfg = load_scene(name, version)
        want = fx[f"{name}-v{version}"]
        assert scene_digest(cfg) == want["cfg_sha256"], f"{name}-v{version}: scene table differs from plb/envs/{name.lower()}.yml"
        x = np.ascontiguousarray(Shapes(cfg.SHAPES).get()[0], np.float64)
        assert len(x) == want["n"] and hashlib.sha256(x.tobytes()).hexdigest() == want["x_sha256"], f"{name}-v{version}: particle cloud"


@pytest.mark.skipif(not os.path.isdir("/root/reference/plb/envs"), reason="the reference checkout is only present in the build container")
def test_scene_fixture_is_what_the_reference_files_say_today():
    """Build container: the committed digests are those of the reference's YAML files as they lie under /root/reference (merged by
    load_variant_file, the counterpart of PlasticineEnv.load_varaints, env.py:63-86)."""
    from tests.util import scene_digest
    from plasticinelab_amd.envs.scenes import ENV_NAMES, load_variant_file
    fx = _scene_fixture()
    assert len(fx) == 50
    for name in ENV_NAMES:
        for version in range(1, 6):
            assert scene_digest(load_variant_file(f"/root/reference/plb/envs/{name.lower()}.yml", version)) == fx[f"{name}-v{version}"]["cfg_sha256"]
