"""Initial particle sampler.

Behavioural mirror of /root/reference/plb/engine/shapes/shape_maker.py:13-76:
same numpy legacy-seed-0 random stream, so "identical Move-v1 initial
conditions" is reproducible bit for bit (checked in tests against a fixture
generated from the reference file itself, tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

from ..config import as_value

_PALETTE = ((127 << 16) + 127, 127 << 8, 127, 127 << 16)


def _rotmat(q):
    w, x, y, z = q
    n = w * w + x * x + y * y + z * z
    s = 2.0 / n
    return np.array([
        [1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
        [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
        [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


class Shapes:
    """Builds the particle cloud from ``cfg.SHAPES`` entries (``box`` / ``sphere``)."""

    dim = 3

    def __init__(self, cfg):
        self.objects, self.colors = [], []
        saved = np.random.get_state()
        np.random.seed(0)                       # shape_maker.py:20-21
        try:
            for entry in cfg:
                kw = {k: as_value(v) for k, v in dict(entry).items() if k != "shape"}
                kind = entry["shape"]
                if kind == "box":
                    self.add_box(**kw)
                elif kind == "sphere":
                    self.add_sphere(**kw)
                else:
                    raise NotImplementedError(f"Shape {kind} is not supported!")
        finally:
            np.random.set_state(saved)

    @staticmethod
    def get_n_particles(volume):                # shape_maker.py:33-34
        return max(int(volume / 0.2 ** 3) * 10000, 1)

    def _push(self, pts, color, init_rot):
        if init_rot is not None:
            c = pts.mean(axis=0)
            pts = (pts[:, :self.dim] - c) @ _rotmat(init_rot).T + c
        self.objects.append(pts[:, :self.dim])
        if color is None or isinstance(color, int):
            value = _PALETTE[len(self.objects) - 1] if color is None else color
            color = np.full(len(pts), value, np.int32)
        self.colors.append(color)

    def add_box(self, init_pos, width, n_particles=10000, color=None, init_rot=None):
        width = np.full(self.dim, width) if isinstance(width, float) else np.asarray(width)
        if n_particles is None:
            n_particles = self.get_n_particles(np.prod(width))
        u = np.random.random((n_particles, self.dim))
        self._push((u * 2 - 1) * (0.5 * width) + np.asarray(init_pos), color, init_rot)

    def add_sphere(self, init_pos, radius, n_particles=10000, color=None, init_rot=None):
        if n_particles is None:
            n_particles = self.get_n_particles(radius ** 3 * 4 * np.pi / 3)
        d = np.random.normal(size=(n_particles, self.dim))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        rad = np.random.random(size=(n_particles, 1)) ** (1.0 / self.dim)
        self._push(d * rad * radius + np.asarray(init_pos)[:self.dim], color, init_rot)

    def get(self):
        assert self.objects, "please add at least one shape into the scene"
        return np.concatenate(self.objects), np.concatenate(self.colors)
