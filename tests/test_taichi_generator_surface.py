"""CPU tier, build container only (skips where /root/reference is absent): every attribute and method name that
tests/golden/make_taichi_golden.py calls on the reference -- env.taichi_env, te.set_state / step / compute_loss, sim.substep /
substep_grad / get_x / x.grad ..., te.primitives.get_grad / set_action, p.position, te.loss.loss / set_weights(keywords) -- exists
in the corresponding class of the reference's source (AST of both sides; nothing of the reference is imported or executed: it
needs Taichi).  The generator can only be RUN where Taichi 0.7.14 is installed; this keeps it from failing there on a typo."""
import ast
import os

import pytest

from tests.util import GOLDEN

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "plb")), reason="the reference checkout is only present in the build container")

# variable (or dotted prefix) in the generator -> (reference file, class)
OBJECTS = {
    "env": ("plb/envs/env.py", "PlasticineEnv"),
    "te": ("plb/engine/taichi_env.py", "TaichiEnv"),
    "sim": ("plb/engine/mpm_simulator.py", "MPMSimulator"),
    "te.primitives": ("plb/engine/primitive/primitives.py", "Primitives"),
    "te.loss": ("plb/engine/losses/loss.py", "Loss"),
    "p": ("plb/engine/primitive/primive_base.py", "Primitive"),
}


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return ".".join(reversed(parts))
    return None


def class_surface(path, cls):
    """names a class of the reference defines: methods / properties, and every `self.name = ...` in its body"""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    node = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls)
    names, params = set(), {}
    for n in ast.walk(node):
        if isinstance(n, ast.FunctionDef):
            names.add(n.name)
            params[n.name] = [a.arg for a in n.args.args[1:]]
        elif isinstance(n, (ast.Assign, ast.AugAssign, ast.AnnAssign)):
            for t in (n.targets if isinstance(n, ast.Assign) else [n.target]):
                for s in ast.walk(t):
                    if isinstance(s, ast.Attribute) and isinstance(s.value, ast.Name) and s.value.id == "self":
                        names.add(s.attr)
    return names, params


def test_every_name_the_generator_uses_exists_in_the_reference():
    gen = ast.parse(open(os.path.join(GOLDEN, "make_taichi_golden.py")).read())
    used = {k: set() for k in OBJECTS}
    for node in ast.walk(gen):
        if not isinstance(node, ast.Attribute):
            continue
        d = dotted(node)
        if d is None:
            continue
        parts = d.split(".")
        for n in range(len(parts) - 1, 0, -1):                   # longest known prefix: te.loss.loss -> Loss.loss, te.loss -> TaichiEnv.loss
            prefix = ".".join(parts[:n])
            if prefix in OBJECTS:
                used[prefix].add(parts[n])
                break
    assert used["sim"] >= {"substep", "substep_grad", "get_x", "get_v", "get_state", "cur", "x", "v", "C", "F"}      # the walk found the calls at all
    assert used["te"] >= {"set_state", "get_state", "step", "compute_loss", "loss", "primitives", "simulator"}
    for var, (path, cls) in OBJECTS.items():
        names, _ = class_surface(path, cls)
        missing = sorted(used[var] - names)
        assert not missing, f"make_taichi_golden.py uses {var}.{missing} but {path}::{cls} defines no such name"


def test_keyword_arguments_and_entry_points_match():
    gen_src = open(os.path.join(GOLDEN, "make_taichi_golden.py")).read()
    gen = ast.parse(gen_src)
    calls = [n for n in ast.walk(gen) if isinstance(n, ast.Call)]
    # te.loss.set_weights(sdf=, density=, contact=, is_soft_contact=)
    _, params = class_surface("plb/engine/losses/loss.py", "Loss")
    sw = [c for c in calls if dotted(c.func) == "te.loss.set_weights"]
    assert sw and all({k.arg for k in c.keywords} <= set(params["set_weights"]) for c in sw), params["set_weights"]
    # te.set_state(state, softness, is_copy), te.primitives.set_action(s, n_substeps, action): positional counts fit
    _, tparams = class_surface("plb/engine/taichi_env.py", "TaichiEnv")
    for c in calls:
        if dotted(c.func) == "te.set_state":
            assert len(c.args) == len(tparams["set_state"]) == 3
    _, pparams = class_surface("plb/engine/primitive/primitives.py", "Primitives")
    for c in calls:
        if dotted(c.func) == "te.primitives.set_action":
            assert len(c.args) == len(pparams["set_action"]) == 3
        if dotted(c.func) == "te.primitives.get_grad":
            assert len(c.args) == len(pparams["get_grad"]) == 1
    # plb.envs.make("Move-v1"): the factory exists and the id is registered under that name
    envs = open(os.path.join(REF, "plb/envs/__init__.py")).read()
    assert "def make(" in envs and "'Move'" in envs.replace('"', "'") and "-v" in envs
    # ti.Tape(loss=te.loss.loss): Loss.loss is a Taichi field with a gradient
    loss_src = open(os.path.join(REF, "plb/engine/losses/loss.py")).read()
    assert "self.loss = ti.field(" in loss_src and "needs_grad=True" in loss_src
    # the seeded adjoints go into sim.x.grad / v.grad / C.grad / F.grad: fields declared with needs_grad
    sim_src = open(os.path.join(REF, "plb/engine/mpm_simulator.py")).read()
    for f in ("self.x", "self.v", "self.C", "self.F"):
        line = next(ln for ln in sim_src.splitlines() if ln.strip().startswith(f + " ="))
        assert "needs_grad=True" in line, line
