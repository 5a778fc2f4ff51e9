"""Configuration tree for the MI355X MPM engine.

Mirrors the key layout of the reference's yacs tree
(/root/reference/plb/config/default_config.py:1-78) so that scene files written
for the reference load unchanged, but is built on plain PyYAML (yacs is not a
dependency here).  ``CfgNode`` is an attribute-access dict; anything exposing
the same attributes (including a real yacs node) is accepted by the engine.
"""
from __future__ import annotations

import ast
import copy
import operator
from typing import Any, Iterable, Mapping, Optional

import yaml


class CfgNode(dict):
    """dict with attribute access, recursive over nested mappings."""

    def __init__(self, init: Optional[Mapping[str, Any]] = None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    # yacs compatibility no-ops (reference callers freeze/defrost the tree)
    def defrost(self):
        return self

    def freeze(self):
        return self

    def merge(self, other: Mapping[str, Any], strict: bool = True) -> "CfgNode":
        """Recursive merge; with ``strict`` unknown keys are an error, which is
        how the reference catches typos in scene files (yacs behaviour)."""
        for k, v in other.items():
            if isinstance(v, Mapping) and isinstance(self.get(k), CfgNode):
                self[k].merge(v, strict)
            else:
                if strict and k not in self:
                    raise KeyError(f"Non-existent config key: {k}")
                self[k] = _wrap(v)
        return self


def _wrap(v):
    if isinstance(v, CfgNode):
        return v
    if isinstance(v, Mapping):
        return CfgNode(v)
    if isinstance(v, list):
        return [_wrap(i) for i in v]
    return v


_BIN_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
            ast.FloorDiv: operator.floordiv, ast.Mod: operator.mod, ast.Pow: operator.pow,
            ast.LShift: operator.lshift, ast.RShift: operator.rshift, ast.BitOr: operator.or_,
            ast.BitAnd: operator.and_, ast.BitXor: operator.xor}
_UN_OPS = {ast.UAdd: operator.pos, ast.USub: operator.neg, ast.Invert: operator.invert}


def _arith(node):
    """Numbers, tuples / lists of them and arithmetic on them -- nothing else (no names, calls, attributes)."""
    if isinstance(node, ast.Expression):
        return _arith(node.body)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float, bool)) or \
            isinstance(node, ast.Constant) and node.value is None:
        return node.value
    if isinstance(node, ast.Tuple):
        return tuple(_arith(e) for e in node.elts)
    if isinstance(node, ast.List):
        return [_arith(e) for e in node.elts]
    if isinstance(node, ast.BinOp) and type(node.op) in _BIN_OPS:
        a, b = _arith(node.left), _arith(node.right)
        # numbers only on either side (no sequence repetition / concatenation), bounded exponents and shift counts, and a
        # bounded result: a scene string such as '1<<(1<<36)' or '(0,)*10**10' must not be able to allocate gigabytes
        if not all(isinstance(t, (int, float, bool)) for t in (a, b)):
            raise ValueError("arithmetic on a sequence")
        if isinstance(node.op, (ast.Pow, ast.LShift, ast.RShift)) and abs(b) > 64:
            raise ValueError("exponent / shift count too large")
        r = _BIN_OPS[type(node.op)](a, b)
        if isinstance(r, int) and abs(r) > 1 << 128:
            raise ValueError("result too large")
        return r
    if isinstance(node, ast.UnaryOp) and type(node.op) in _UN_OPS:
        return _UN_OPS[type(node.op)](_arith(node.operand))
    raise ValueError(f"not plain arithmetic: {ast.dump(node)}")


def as_value(v):
    """Scene files carry tuples and arithmetic as strings, e.g. ``(0.5, 0.5, 0.5)``
    or ``0.2049/2`` or ``127<<16``; the reference ``eval``s them
    (shape_maker.py:23).  Here they go through a small arithmetic evaluator over the
    parsed expression (numbers, tuples, + - * / // % ** << >> | & ^): a scene file
    cannot run code.  Anything else is returned unchanged, as a string."""
    if isinstance(v, str):
        try:
            return _arith(ast.parse(v.strip(), mode="eval"))
        except Exception:
            return v
    return v


def get_cfg_defaults() -> CfgNode:
    """Same keys and defaults as reference default_config.py:12-78."""
    return CfgNode({
        "SIMULATOR": {
            "dim": 3, "quality": 1, "yield_stress": 50.0, "dtype": "float64",
            "max_steps": 1024, "n_particles": 9000, "E": 5e3, "nu": 0.2,
            "ground_friction": 1.5, "gravity": (0, -1, 0),
        },
        "PRIMITIVES": [],
        "SHAPES": [],
        "RENDERER": {
            "spp": 50, "max_ray_depth": 2, "image_res": (512, 512),
            "voxel_res": (168, 168, 168), "target_res": (64, 64, 64),
            "dx": 1.0 / 150, "sdf_threshold": 0.37 * 0.56, "bake_size": 6,
            "use_roulette": False, "light_direction": (2.0, 1.0, 0.7),
            "camera_pos": (0.5, 1.2, 4.0), "camera_rot": (0.2, 0),
            "use_directional_light": False, "max_num_particles": 1000000,
        },
        "ENV": {
            "loss": {"soft_contact": False,
                     "weight": {"sdf": 10, "density": 10, "contact": 1},
                     "target_path": ""},
            "n_observed_particles": 200,
        },
        "VARIANTS": [],
    })


def load(path: Optional[str] = None, opts: Optional[Iterable] = None) -> CfgNode:
    """reference plb/config/utils.py:33-40."""
    cfg = get_cfg_defaults()
    if path is not None:
        with open(path) as f:
            cfg.merge(yaml.safe_load(f) or {}, strict=True)
    if opts:
        opts = list(opts)
        for k, v in zip(opts[0::2], opts[1::2]):
            node = cfg
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _wrap(v)
    return cfg


def merge_dict(a, b):
    """reference plb/envs/utils.py:3-19 -- b overrides a; unknown keys rejected."""
    if b is None:
        return a
    a = copy.deepcopy(a)
    for key in b:
        if key not in a:
            raise ValueError("Key is not in dict A!")
        if isinstance(b[key], Mapping):
            a[key] = merge_dict(a[key], b[key])
        else:
            a[key] = b[key]
    return a


def merge_lists(a, b):
    """reference plb/envs/utils.py:22-31 -- positional merge of variant lists."""
    out = []
    for i, x in enumerate(a):
        out.append(merge_dict(x, b[i]) if i < len(b) else x)
    return out
